"""Single-stream decode step time as a function of the context length already in the KV cache."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import DecodeEngine
dev = torch.device("cuda:0")
cfg = OneBitLlamaConfig.llama_7b()
model = build_synthetic_model(cfg, seed=1, device=dev)
eng = DecodeEngine(model, max_len=2048, attn_splits=int(os.environ.get("SPLITS", "8")), long_context_from=int(os.environ.get("LONG_FROM", "384")),
                   long_attention=os.environ.get("LONG_ATTN", "keyblock"), attn_chunk=int(os.environ.get("CHUNK", "128")))
for kc, vc in eng.cache.layers:                       # plausible cache contents
    kc.normal_(); vc.normal_()
for ctx in (16, 128, 256, 512, 1024, 1900):
    eng.set_state(5, ctx)
    for _ in range(4): eng.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 32
    for _ in range(n): eng.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("context %4d..%4d: %.3f ms/token  %.0f tok/s" % (ctx + 4, ctx + 4 + n, dt * 1e3, 1 / dt))
