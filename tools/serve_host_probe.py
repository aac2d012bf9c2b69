#!/usr/bin/env python3
"""Host time per decode-only step of ContinuousBatcher (every step ends in one host sync for the tokens, so host work serialises with the GPU):
cProfile of a request stream on 7B shapes.  python tools/serve_host_probe.py"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onebit_amd.llama import build_synthetic_model
from onebit_amd.serving import ContinuousBatcher
dev = torch.device("cuda:0")
cfg = bench.model_config(os.environ.get("MODEL", "7b"))
model = build_synthetic_model(cfg, seed=4242, device=dev)
g = torch.Generator().manual_seed(11)
prompts = [torch.randint(0, cfg.vocab_size, (int(n),), generator=g).tolist() for n in torch.randint(64, 513, (64,), generator=g)]
cb = ContinuousBatcher(model, max_batch=32, max_len=640, prefill_chunk=512, max_step_tokens=4128)
for p in prompts[:4]:
    cb.add_request(p, 4)
cb.run()
cb.steps = cb.mixed_steps = cb.graph_steps = 0
cb.time_mixed = cb.time_decode = 0.0
for p in prompts:
    cb.add_request(p, 64)
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
cb.run()
pr.disable()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%.3f s, %d steps (%d decode-only graph steps at %.3f ms each incl. host, %d mixed at %.2f ms)" % (
    dt, cb.steps, cb.graph_steps, cb.time_decode / max(cb.graph_steps, 1) * 1e3, cb.mixed_steps, cb.time_mixed / max(cb.mixed_steps, 1) * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:34]))
