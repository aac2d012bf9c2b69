"""Per-kernel times of the 32-slot batched decode step (graph replay) from torch.profiler (roctracer), no rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.serving import ContinuousBatcher
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "7b"
cfg = OneBitLlamaConfig.llama_13b() if name == "13b" else OneBitLlamaConfig.llama_7b()
cfg.num_hidden_layers = 8
model = build_synthetic_model(cfg, seed=1, device=dev)
g = torch.Generator().manual_seed(0)
cb = ContinuousBatcher(model, max_batch=32, max_len=160)
for _ in range(32):
    cb.add_request(torch.randint(0, cfg.vocab_size, (16,), generator=g).tolist(), 100)
for _ in range(6):
    cb.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        cb.step()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total / max(e.count, 1)) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1] * r[2])
tot = sum(r[1] * r[2] for r in rows)
for k, c, t in rows[:14]:
    print("%-110s calls %5d avg %8.2f us  %5.1f%%" % (k[:110], c, t, 100 * c * t / tot))
print("device time per step (8 layers): %.1f us" % (tot / 10))
