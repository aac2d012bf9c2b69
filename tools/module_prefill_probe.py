#!/usr/bin/env python3
"""The module path of a prefill call -- BitLinearInf.forward on [8 x 2048, K] rows (BASELINE configs[2]) -- timed per call with HIP events
(median / min of 20 after warm-up), for the three 7B projection shapes.  OB_LN_ROWS=0: the previous LayerNorm pass (A/B).
python tools/module_prefill_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
T = 8 * 2048
g = torch.Generator(device=dev).manual_seed(77)
for K, N, bias in ((4096, 11008, False), (11008, 4096, False), (4096, 4096, False), (4096, 4096, True)):
    m = BitLinearInf(K, N, bias=bias, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8, device=dev).view(torch.int8)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g, device=dev))).half()
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g, device=dev))).half()
    if bias:
        m.bias.data = torch.randn(N, generator=g, device=dev).half()
    x = torch.randn(T, K, generator=g, device=dev).half()
    dt, dmin = bench._timed(lambda: m(x), dev, 1)
    fl = 2.0 * T * K * N
    print("%5d -> %5d%s: %.3f ms (min %.3f)  %.0f TFLOP/s (best %.0f)" % (K, N, " +bias" if bias else "", dt * 1e3, dmin * 1e3, fl / dt / 1e12, fl / dmin / 1e12), flush=True)
