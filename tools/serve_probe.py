"""Continuous-batching throughput (BASELINE config 5 shape: 32 sequences, mixed prefill + decode)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.serving import ContinuousBatcher
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
cfg = OneBitLlamaConfig.llama_13b() if name == "13b" else OneBitLlamaConfig.llama_7b()
model = build_synthetic_model(cfg, seed=1, device=dev)
g = torch.Generator().manual_seed(0)
for nreq, new in ((32, 32), (8, 32), (1, 32)):
    cb = ContinuousBatcher(model, max_batch=32, max_len=640)
    lens = [512] * 8 + [16] * 24 if nreq == 32 else [16] * nreq       # 8 prefills of 512 + 24 short prompts
    for n in lens:
        cb.add_request(torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist(), new)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = cb.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    gen = sum(len(v) for v in out.values())
    print("%s: %2d requests, %d steps, %d tokens scheduled (%d generated) in %.3f s -> %.1f generated tok/s, %.1f scheduled tok/s"
          % (name, nreq, cb.steps, cb.tokens_scheduled, gen, dt, gen / dt, cb.tokens_scheduled / dt))
