#!/usr/bin/env python3
"""Copies the summaries tools/r05_profile.sh left under gpurun_out/r05prof into profiles/r05_*, keeping the '#' header lines the
committed files carry (what the pass was), and restamps profiles/pmc_traffic.json.  Usage: python tools/refresh_profiles.py"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r05prof")
MAP = {"bench_profiled.json": "r05_bench_profiled_run.json", "bench_kernel_stats.csv": "r05_bench_rocprofv3_kernel_stats.csv",
       "pmc_FETCH_SIZE.txt": "r05_pmc_FETCH_SIZE.txt", "pmc_decode_issue.txt": "r05_pmc_decode_issue.txt",
       "decode_insitu.txt": "r05_decode_insitu.txt", "serve_kernels.txt": "r05_serve_step_kernels.txt",
       "prefill_layer_kernel_stats.csv": "r05_prefill_layer_rocprofv3_kernel_stats.csv", "pmc_prefill_mfma.txt": "r05_pmc_prefill_mfma.txt",
       "persist_skeleton.txt": "r05_persist_skeleton_run3.txt"}
for src, dst in MAP.items():
    sp, dp = os.path.join(SRC, src), os.path.join(ROOT, "profiles", dst)
    if not os.path.exists(sp):
        print("missing", sp)
        continue
    header = []
    if os.path.exists(dp) and not dst.endswith((".csv", ".json")):
        for line in open(dp):
            if not line.startswith("#"):
                break
            header.append(line)
    body = open(sp).read()
    if dst.endswith(".json"):
        body = "\n".join(l for l in body.splitlines() if l.startswith("{")) + "\n"
    open(dp, "w").write("".join(header) + body)
    print("wrote", dst, len(header), "header lines")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_pmc_traffic.py"), os.path.join(SRC, "pmc_FETCH_SIZE.txt")])
