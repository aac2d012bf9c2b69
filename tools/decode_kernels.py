"""Per-kernel times of the single-sequence decode step IN SITU (HIP-graph replay of onebit_decode_step on the 7B
synthetic checkpoint) from torch.profiler (roctracer): what each launch costs inside the real chain, as opposed to the
isolated per-kind chains of bench.py.  Usage: python tools/decode_kernels.py [13b]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import DecodeEngine
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
cfg = OneBitLlamaConfig.llama_13b() if name == "13b" else OneBitLlamaConfig.llama_7b()
model = build_synthetic_model(cfg, seed=1, device=dev)
eng = DecodeEngine(model, max_len=256)
eng.prefill(torch.randint(0, cfg.vocab_size, (1, 16), generator=torch.Generator().manual_seed(0)).to(dev))
for _ in range(24):
    eng.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(64):
    eng.step()
e1.record(); torch.cuda.synchronize()
print("%s: %.4f ms/token unprofiled (%.1f tok/s)" % (name, e0.elapsed_time(e1) / 64, 64e3 / e0.elapsed_time(e1)))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(16):
        eng.step()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total / max(e.count, 1)) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1] * r[2])
tot = sum(r[1] * r[2] for r in rows)
for k, c, t in rows[:12]:
    print("%-100s calls %5d avg %8.2f us  %5.1f%%" % (k[:100], c, t, 100 * c * t / tot))
print("device time per token: %.1f us (kernel time only, gaps excluded)" % (tot / 16))
