"""Serving loop A/B: ContinuousBatcher with one host synchronisation per step (max_burst=1) against decode bursts
(max_burst=16), same requests, same positions; ms per steady-state step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.serving import ContinuousBatcher
dev = torch.device("cuda:0")
cfg = OneBitLlamaConfig.llama_7b()
model = build_synthetic_model(cfg, seed=1, device=dev)
for rep in range(2):
    for mb in (1, 16, 4):
        g = torch.Generator().manual_seed(0)
        cb = ContinuousBatcher(model, max_batch=32, max_len=160, max_burst=mb)
        for _ in range(32):
            cb.add_request(torch.randint(0, cfg.vocab_size, (16,), generator=g).tolist(), 120)
        while cb.steps < 20:
            cb.step()
        torch.cuda.synchronize(); s0 = cb.steps; t0 = time.perf_counter()
        while cb.steps - s0 < 48:
            cb.step()
        torch.cuda.synchronize(); n = cb.steps - s0
        print("max_burst %2d: %.3f ms/step over %d steps (positions %d..%d)" % (mb, (time.perf_counter() - t0) / n * 1e3, n, 16 + s0, 16 + cb.steps), flush=True)
        del cb
