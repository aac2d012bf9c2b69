#!/usr/bin/env python3
"""The fused K-sharded decode step (config 4) at N = 1 against the cached context: one workgroup per head (attn_chunk 0) vs the key-block
attention route (onebit_kshard_state_t.attn_chunk).  13B shapes, synthetic weights, one HIP graph per decoder.
python tools/kshard_ctx_probe.py [max_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd.llama import build_synthetic_model
from onebit_amd.sharded import FusedKShardedDecoder
dev = torch.device("cuda:0")
max_len = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = bench.model_config(os.environ.get("MODEL", "13b"))
cfg.max_position_embeddings = max(cfg.max_position_embeddings, max_len)
model = build_synthetic_model(cfg, seed=4242, device=dev)
for ac in (0, 128, 256):
    dec = FusedKShardedDecoder(model, 0, 1, max_len=max_len, attn_chunk=ac)
    for k, v in dec.cache.layers:
        k.normal_(0, 0.5); v.normal_(0, 0.5)
    row = []
    for ctx in (16, 128, 512, 1024, max_len - 40):
        dec.set_state(5, ctx)
        dec.capture()
        for _ in range(3):
            dec.set_state(5, ctx); dec.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 24
        dec.set_state(5, ctx)
        for _ in range(n):
            dec.step()
        torch.cuda.synchronize()
        row.append("ctx %4d..%4d: %.3f ms" % (ctx, ctx + n, (time.perf_counter() - t0) / n * 1e3))
    print("attn_chunk %3d (%s):  " % (ac, "one workgroup per head" if ac == 0 else "%d splits" % dec.backend._state.attn_splits) + "   ".join(row), flush=True)
    del dec
