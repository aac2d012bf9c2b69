#!/usr/bin/env python3
"""BitLinearInf.forward (the module path: onebit_linear_forward, LayerNorm included) against the number of token rows, 7B shapes:
ms per call and TFLOP/s -- where the route changes (GEMV / skinny / 16-token tiles / 128 x 128 / LDS-DMA GEMM) shows as steps.
python tools/module_T_sweep.py [13b]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
H, I = (5120, 13824) if len(sys.argv) > 1 and sys.argv[1] == "13b" else (4096, 11008)
g = torch.Generator(device=dev).manual_seed(77)
Ts = [int(t) for t in os.environ.get("TS", "32,64,65,96,128,192,256,320,384,512,640,768,1024,1280,1536,2048,4096").split(",")]
for K, N in ((H, H), (H, I), (I, H)):
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8, device=dev).view(torch.int8)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g, device=dev))).half()
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g, device=dev))).half()
    row = []
    for T in Ts:
        x = torch.randn(T, K, generator=g, device=dev).half()
        dt, dmin = bench._timed(lambda: m(x), dev, 1, warm=3, iters=10)
        row.append("%d: %.0f us %.0f TF" % (T, dt * 1e6, 2.0 * T * K * N / dt / 1e12))
    print("%5d -> %5d   " % (K, N) + " | ".join(row), flush=True)
