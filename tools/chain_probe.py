"""Batched decode step (config 5) as 1 / 2 / 4 independent chains (onebit_batch_state_t.chains): ms per steady-state
step under HIP-graph replay, one model build per size.  Usage: python tools/chain_probe.py [7b] [13b] [slots=32]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import BatchedDecodeStep
dev = torch.device("cuda:0")
names = [a for a in sys.argv[1:] if a in ("7b", "13b")] or ["7b"]
slots = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("slots=")] or [32]
for name in names:
    cfg = OneBitLlamaConfig.llama_13b() if name == "13b" else OneBitLlamaConfig.llama_7b()
    model = build_synthetic_model(cfg, seed=1, device=dev)
    for B in slots:
        max_len = 160
        ref_tokens = None
        for chains in (1, 2, 4, 1, 2):
            cache = model.new_cache(B, max_len)
            step = BatchedDecodeStep(model, cache.layers, B, max_len, sample=True, chains=chains)
            g0 = torch.Generator().manual_seed(0)
            step.tokens.copy_(torch.randint(0, cfg.vocab_size, (B,), generator=g0, dtype=torch.int32))
            step.pos.fill_(16)
            step.launch(); torch.cuda.synchronize()
            toks = step.next_tokens.cpu().clone()
            if ref_tokens is None: ref_tokens = toks
            same = bool((toks == ref_tokens).all())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step.launch()
            for _ in range(5): gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n): gr.replay()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print("%s  %2d slots  chains %d: %.3f ms/step -> %.0f tok/s   (tokens equal to 1 chain: %s)" % (name, B, chains, ms, B / ms * 1e3, same), flush=True)
            del gr, step, cache
    del model
    torch.cuda.empty_cache()
