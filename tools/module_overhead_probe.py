#!/usr/bin/env python3
"""Host-side cost of one BitLinearInf.forward call (the drop-in module path): wall clock per call over a long loop at T = 1 (the GPU work is a
~5 us GEMV + LayerNorm: the loop is host-bound) and the cProfile breakdown.  python tools/module_overhead_probe.py"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
K = N = 4096
m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8, device=dev).view(torch.int8)
for T in (1, 32, 256):
    x = torch.randn(T, K, generator=g, device=dev).half()
    for _ in range(200):
        m(x)
    torch.cuda.synchronize()
    n = 5000
    t0 = time.perf_counter()
    for _ in range(n):
        m(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("T = %3d: %.2f us per call to enqueue, %.2f us per call including the drain" % (T, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
x = torch.randn(1, K, generator=g, device=dev).half()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    m(x)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:34]))
