#!/usr/bin/env python3
"""Time to first token of ONE prompt, 7B (or 13b) shapes: DecodeEngine.prefill (module path with the fused glue, logits of every row) vs one
onebit_mixed_step over the same rows (MixedStep: lm_head on the last row only).  python tools/ttft_probe.py [13b]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd.llama import build_synthetic_model
from onebit_amd.engine import DecodeEngine, MixedStep
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
cfg = bench.model_config(name)
model = build_synthetic_model(cfg, seed=4242, device=dev)
max_len = 2048
eng = DecodeEngine(model, max_len=max_len)
ms = MixedStep(model, eng.cache.layers, 1, max_len, max_rows=2048)
g = torch.Generator().manual_seed(3)
def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
print("%s: prompt tokens: DecodeEngine.prefill ms | onebit_mixed_step ms" % name)
for S in [int(s) for s in os.environ.get("SS", "16,64,65,100,128,200,300,400,512,700,1000,1500,2040").split(",")]:
    ids = torch.randint(0, cfg.vocab_size, (1, S), generator=g).to(dev)
    toks = ids[0].tolist()
    a = timed(lambda: eng.prefill(ids))
    first = eng.first_token
    b = timed(lambda: ms.launch([(0, 0, toks)]))
    nxt = int(ms.launch([(0, 0, toks)])[0])
    print("%5d: %8.2f | %8.2f   first token %s" % (S, a, b, "equal" if nxt == first else "%d vs %d" % (first, nxt)), flush=True)
