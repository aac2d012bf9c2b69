#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a `rocprofv3 --pmc FETCH_SIZE` summary (tools/pmc_summary.py output),
stamped with the hash of the kernel sources so bench.py only reports it for the code it was taken on.
Usage: python tools/make_pmc_traffic.py gpurun_out/.../pmc_FETCH_SIZE.txt"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha
# template arguments: KV, MS, ALIGNED, PRO, MATH, NPROJ, PST, WGP, BIAS, ZOUT (round 4: q|k|v and gate|up are one-projection-per-workgroup
# launches, o_proj runs on the integer path)
# round 5: two more template arguments (BIAS, ZOUT), both false in the product step of a bias-free checkpoint
names = {"<1, 4, true, 2, 1, 1, true, true, false, false>": "qkv", "<1, 1, true, 0, 1, 1, false, false, false, false>": "o",
         "<1, 6, true, 2, 1, 1, true, true, false, false>": "gate_up", "<3, 1, true, 3, 1, 1, true, false, false, false>": "down"}
out = {}
for line in open(sys.argv[1]):
    m = re.search(r"FETCH_SIZE avg ([0-9.]+)", line)
    if not m:
        continue
    for key, nm in names.items():
        if "ob_dec_gemv_kernel" + key in line:
            out[nm] = int(float(m.group(1)) * 1024 * 2)
    if "ob_dec_lmhead_kernel" in line:
        out["lm_head"] = int(float(m.group(1)) * 1024 * 2)
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE (own pass) -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-prefill "
                     "--no-serve --no-roofline; per-kernel averages via tools/pmc_summary.py",
           "units": "FETCH_SIZE is reported in KiB; on gfx950 it counts 128-B fabric requests as 64 B for wide coalesced streams "
                    "(MI355X_MICROARCH.md, HBM section), so bytes = FETCH_SIZE * 1024 * 2",
           "csrc_sha": csrc_sha(), "bytes_per_launch": out}, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(out)
