"""Continuous-batching throughput (BASELINE config 5 shape: 32 sequences, mixed prefill + decode).
Part 1: whole job (8 x 512-token + 24 x 16-token prompts, 32 new tokens each, prefill and graph capture included).
Part 2: steady-state decode step (all slots decoding), HIP-graph replay only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.serving import ContinuousBatcher
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
cfg = OneBitLlamaConfig.llama_13b() if name == "13b" else OneBitLlamaConfig.llama_7b()
model = build_synthetic_model(cfg, seed=1, device=dev)
g = torch.Generator().manual_seed(0)
rnd = lambda n: torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist()
STEADY_ONLY = bool(os.environ.get("OB_STEADY_ONLY"))           # profiling: just the 32-slot steady-state decode step
if not STEADY_ONLY:
    cb = ContinuousBatcher(model, max_batch=32, max_len=640)
    for n in [512] * 8 + [16] * 24:
        cb.add_request(rnd(n), 32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = cb.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    gen = sum(len(v) for v in out.values())
    print("%s job: 32 requests, %d steps (%d graph), %d tokens scheduled (%d generated) in %.3f s -> %.1f generated tok/s"
          % (name, cb.steps, cb.graph_steps, cb.tokens_scheduled, gen, dt, gen / dt))
for nslots, max_len in (((32, 160),) if STEADY_ONLY else ((32, 160), (32, 640), (8, 160))):
    cb = ContinuousBatcher(model, max_batch=nslots, max_len=max_len)
    for _ in range(nslots):
        cb.add_request(rnd(16), 100)
    for _ in range(6):
        cb.step()                                   # prefill step, capture, warm replays
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        cb.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%s steady decode: %2d slots, max_len %3d: %.2f ms/step -> %.0f tok/s aggregate" % (name, nslots, max_len, dt * 1e3, nslots / dt))
