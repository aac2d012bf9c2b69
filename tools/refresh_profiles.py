#!/usr/bin/env python3
"""Copies the summaries tools/r04_profile.sh left under gpurun_out/r04prof into profiles/r04_*, keeping the '#' header lines the
committed files carry (what the pass was), and restamps profiles/pmc_traffic.json.  Usage: python tools/refresh_profiles.py"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r04prof")
MAP = {"bench_profiled.json": "r04_bench_profiled_run.json", "bench_kernel_stats.csv": "r04_bench_rocprofv3_kernel_stats.csv",
       "pmc_FETCH_SIZE.txt": "r04_pmc_FETCH_SIZE.txt", "pmc_decode_issue.txt": "r04_pmc_decode_issue.txt",
       "decode_insitu.txt": "r04_decode_insitu.txt", "serve_kernels.txt": "r04_serve_step_kernels.txt",
       "prefill_layer_kernel_stats.csv": "r04_prefill_layer_rocprofv3_kernel_stats.csv", "pmc_prefill_mfma.txt": "r04_pmc_prefill_mfma.txt",
       "attn.txt": "r04_attention_probe.txt", "pmc_attention.txt": "r04_pmc_attention.txt"}
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
