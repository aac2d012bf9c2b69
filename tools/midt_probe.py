#!/usr/bin/env python3
"""A mid-size mixed step (chunked prefill with a small step budget): 31 decode rows + one 512-token prompt chunk = 543 rows on 13B shapes.
python tools/midt_probe.py [prompt tokens]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd.llama import build_synthetic_model
dev = torch.device("cuda:0")
model = build_synthetic_model(bench.model_config(os.environ.get("MODEL", "13b")), seed=4242, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
r = bench.measure_mixed_step(model, dev, slots=32, n_prefill=1, prompt=n, ctx=128, iters=8, requests=8, new_tokens=4)
print(json.dumps(r["step"]))
