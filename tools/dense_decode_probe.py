#!/usr/bin/env python3
"""Context for the decode headline: the SAME model in dense fp16 -- what the reference's layer costs after its unpack, or a plain LLaMA-7B --
as 224 vendor-library GEMVs per token (F.linear at T = 1, hipBLASLt / rocBLAS), distinct weights per layer (12.9 GB), one HIP graph per token,
nothing else (no norms, no attention, no lm_head: flatters the dense path).  python tools/dense_decode_probe.py"""
import time, torch
dev = torch.device("cuda:0")
H, I, L = 4096, 11008, 32
shapes = [(H, H)] * 4 + [(H, I)] * 2 + [(I, H)]
ws = [[torch.randn(n, k, device=dev, dtype=torch.float16) * 0.02 for (k, n) in shapes] for _ in range(L)]
xs = {H: torch.randn(1, H, device=dev, dtype=torch.float16), I: torch.randn(1, I, device=dev, dtype=torch.float16)}
def token():
    for l in range(L):
        for w in ws[l]:
            torch.nn.functional.linear(xs[w.shape[1]], w)
for _ in range(3): token()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    token()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 50
for _ in range(n): g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
nbytes = sum(w.numel() * 2 for l in ws for w in l)
print("dense fp16 7B projections, 224 GEMVs per token under one HIP graph: %.3f ms / token = %.0f tokens/s (weights %.2f GB -> %.2f TB/s; HBM roofline of "
      "these bytes at 8 TB/s: %.3f ms = %.0f tokens/s)" % (dt * 1e3, 1 / dt, nbytes / 1e9, nbytes / dt / 1e12, nbytes / 8e12 * 1e3, 8e12 / nbytes))
