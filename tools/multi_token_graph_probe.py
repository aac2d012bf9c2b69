#!/usr/bin/env python3
"""Does a HIP graph replay cost anything per token?  The decode step's graph holds ONE token (162 kernel nodes); this probe captures k
consecutive steps in one graph (the token / position feedback is on the device) and times tokens at 64..192 cached tokens.
python tools/multi_token_graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd.llama import build_synthetic_model
from onebit_amd.engine import DecodeEngine
dev = torch.device("cuda:0")
model = build_synthetic_model(bench.model_config("7b"), seed=4242, device=dev)
eng = DecodeEngine(model, max_len=512, long_context_from=0)
ids = torch.randint(0, 32000, (1, 64)).to(dev)
eng.prime(ids)
for k in (1, 2, 4, 8):
    g = torch.cuda.CUDAGraph()
    eng.set_state(5, 64)
    eng._launch(); torch.cuda.synchronize()
    eng.set_state(5, 64)
    with torch.cuda.graph(g):
        for _ in range(k):
            eng._launch()
    best = 1e9
    for rep in range(5):
        eng.set_state(5, 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(128 // k):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (128 // k * k))
    print("%d token(s) per graph: %.4f ms / token (%.1f tok/s)" % (k, best * 1e3, 1.0 / best), flush=True)
