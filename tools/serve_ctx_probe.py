#!/usr/bin/env python3
"""Batched decode step (onebit_decode_step_batched, 32 slots, HIP graph) against the cached context length: the one-workgroup-per-
(head, slot) attention vs the key-block form (attn_splits) at several chunk sizes.  python tools/serve_ctx_probe.py [7b|13b]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from onebit_amd.engine import BatchedDecodeStep
from onebit_amd.llama import build_synthetic_model

name = sys.argv[1] if len(sys.argv) > 1 else "7b"
dev = torch.device("cuda:0")
cfg = bench.model_config(name)
model = build_synthetic_model(cfg, seed=7, device=dev)
B, max_len = 32, 2048
caches = model.new_cache(B, max_len).layers
for kc, vc in caches:
    kc.normal_(); vc.normal_()


def timed(step, ctx, iters=20):
    step.tokens.fill_(5)
    step.pos.fill_(ctx - 1)
    step.launch(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step.launch()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


H, D, L = cfg.num_attention_heads, cfg.head_dim, cfg.num_hidden_layers
print("%s, %d slots: ms per step (KV bytes per step at that context in GB; attention share vs the ctx-16 step)" % (name, B))
short = BatchedDecodeStep(model, caches, B, max_len)
base = None
for ctx in (16, 64, 128, 256, 512, 1024, 2040):
    kv_gb = B * H * ctx * D * 2 * 2 * L / 1e9
    t_short = timed(short, ctx)
    row = ["ctx %5d  KV %6.2f GB  one-wg %7.3f" % (ctx, kv_gb, t_short)]
    for chunk in (128, 256, 512):
        ns = max(1, -(-ctx // chunk))
        st = BatchedDecodeStep(model, caches, B, max_len, attn_splits=ns, attn_chunk=chunk)
        t = timed(st, ctx)
        if base is None:
            base = t
        row.append("chunk %d x %2d: %7.3f" % (chunk, ns, t))
        if chunk == 256:
            extra = max(t - base, 1e-6)
            row.append("(KV at %.2f TB/s over the ctx-16 step)" % (kv_gb / extra))
        del st
    print("  ".join(row), flush=True)
