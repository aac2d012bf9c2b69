#!/usr/bin/env python3
"""Request stream through ContinuousBatcher (bench.py's `continuous_batch.mixed.request_stream` workload: 128 requests, prompts 64..512, 64
new tokens, 32 slots, all submitted at t = 0) against the scheduler's knobs: prefill_chunk x max_step_tokens.  python tools/stream_probe.py [13b|7b]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from onebit_amd.llama import build_synthetic_model
from onebit_amd.serving import ContinuousBatcher

name = sys.argv[1] if len(sys.argv) > 1 else "13b"
dev = torch.device("cuda:0")
cfg = bench.model_config(name)
model = build_synthetic_model(cfg, seed=4242, device=dev)
slots, new_tokens, requests = 32, 64, 128
g = torch.Generator(device="cpu").manual_seed(11)
lens = torch.randint(64, 513, (requests,), generator=g).tolist()
prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in lens]
pct = lambda v, q: v[min(len(v) - 1, int(q * len(v)))] * 1e3
print("%s, %d requests, prompts 64..512 (mean %.0f), %d new tokens, %d slots" % (name, requests, sum(lens) / len(lens), new_tokens, slots))
print("%-14s %-16s %9s %9s %7s %7s %9s %9s %9s %9s" % ("prefill_chunk", "max_step_tokens", "gen tok/s", "all tok/s", "mixed", "decode", "TTFT p50", "TTFT p99", "ITL p50", "ITL p99"))
for chunk, budget in ((None, None), (512, 4128), (512, 2080), (256, 2080), (256, 1056), (128, 1056), (128, 544), (64, 288)):
    cb = ContinuousBatcher(model, max_batch=slots, max_len=512 + new_tokens + 64, prefill_chunk=chunk, max_step_tokens=budget)
    for p in prompts[:4]:
        cb.add_request(p, 4)
    cb.run()
    cb.steps = cb.mixed_steps = cb.graph_steps = 0
    cb.time_mixed = cb.time_decode = 0.0
    for p in prompts:
        cb.add_request(p, new_tokens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = cb.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    reqs = [r for r in cb.sched.finished.values() if r.max_new_tokens == new_tokens]
    gen = sum(len(r.out) for r in reqs)
    ttft = sorted(r.t_first - r.t_add for r in reqs)
    itl = sorted((r.t_done - r.t_first) / max(len(r.out) - 1, 1) for r in reqs)
    print("%-14s %-16s %9.0f %9.0f %7d %7d %8.0f ms %7.0f ms %7.2f ms %7.2f ms" % (chunk, budget, gen / dt, (gen + sum(lens)) / dt, cb.mixed_steps, cb.graph_steps,
                                                                          pct(ttft, 0.5), pct(ttft, 0.99), pct(itl, 0.5), pct(itl, 0.99)), flush=True)
    del cb
    torch.cuda.empty_cache()
