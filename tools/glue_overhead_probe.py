#!/usr/bin/env python3
"""Host cost of the fused prefill route (OneBitLlamaForCausalLM.forward with set_fused_glue / set_attention("hip")) at a prompt so short
that the GPU work is negligible: wall clock per forward and the cProfile breakdown.  python tools/glue_overhead_probe.py"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd.llama import build_synthetic_model
dev = torch.device("cuda:0")
model = build_synthetic_model(bench.model_config("7b"), seed=1, device=dev)
model.set_attention("hip").set_fused_glue(True)
for S in (16, 512, 2048):
    ids = torch.randint(0, 32000, (1, S)).to(dev)
    with torch.no_grad():
        for _ in range(3):
            model(ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model(ids)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("S = %4d: %.2f ms per forward to enqueue, %.2f ms including the drain" % (S, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3), flush=True)
ids = torch.randint(0, 32000, (1, 16)).to(dev)
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(20):
        model(ids)
    pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[:40]))
