#!/usr/bin/env python3
"""Instruction-count meter for the decode GEMV kernels: compiles the stamps build to ISA and reports,
per kernel and per phase (segments between the s_memtime stamps), how many instructions a wave
executes straight-line.  The launches are instruction-issue bound, so this is
the local proxy for kernel time.  Usage: python tools/icount.py [extra hipcc flags]"""
import os, subprocess, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/ob_icount.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                       "-DOB_PROFILE_STAMPS", *sys.argv[1:], "-S", "--cuda-device-only", "-o", out,
                       os.path.join(ROOT, "onebit_amd/csrc/onebit_hip.hip")], stderr=subprocess.DEVNULL)
text = open(out).read().split("\n")
# template arguments: KV, MS, ALIGNED, PRO, MATH, NPROJ, PST, WGP
kernels = {"gate|up (WGP, MS 6)": "_Z18ob_dec_gemv_kernelILi1ELi6ELb1ELi2ELi1ELi1ELb1ELb1EEv10ObGemvArgs",
           "q|k|v (WGP, MS 4)": "_Z18ob_dec_gemv_kernelILi1ELi4ELb1ELi2ELi1ELi1ELb1ELb1EEv10ObGemvArgs",
           "gate|up (per-slot)": "_Z18ob_dec_gemv_kernelILi1ELi3ELb1ELi2ELi1ELi2ELb1ELb0EEv10ObGemvArgs",
           "q|k|v (per-slot)": "_Z18ob_dec_gemv_kernelILi1ELi1ELb1ELi2ELi1ELi3ELb1ELb0EEv10ObGemvArgs",
           "down": "_Z18ob_dec_gemv_kernelILi3ELi1ELb1ELi3ELi1ELi1ELb1ELb0EEv10ObGemvArgs",
           "o": "_Z18ob_dec_gemv_kernelILi1ELi1ELb1ELi0ELi0ELi1ELb0ELb0EEv10ObGemvArgs"}
names = ["kernarg", "head", "LN stats", "RMS", "x", "amax", "digits", "1st MFMA", "MFMA", "reduce", "finish", "flush"]
for label, sym in kernels.items():
    try:
        i0 = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
    except StopIteration:
        print(label, "not found"); continue
    body = []
    for l in text[i0 + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"): break
        if not t or t.startswith(";") or t.startswith("."): continue
        body.append(t)
    seg, cur = [], []
    for t in body:
        if t.startswith("s_memtime"): seg.append(cur); cur = []
        else: cur.append(t)
    seg.append(cur)
    tot = sum(len(x) for x in seg)
    print("%-20s total %4d | " % (label, tot) + "  ".join("%s %d" % (names[i] if i < len(names) else "s%d" % i, len(x)) for i, x in enumerate(seg)))
