"""Per-kernel times of the whole-model fused prefill (7B, 8 x 2048 tokens) from torch.profiler."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg = OneBitLlamaConfig.llama_7b()
cfg.num_hidden_layers = int(os.environ.get("LAYERS", "8"))
model = build_synthetic_model(cfg, seed=1, device=dev)
ids = torch.randint(0, cfg.vocab_size, (8, 2048), generator=torch.Generator().manual_seed(5)).to(dev)
model.set_attention(os.environ.get("ATTN", "hip")).set_fused_glue(True)
with torch.no_grad():
    for _ in range(2): model(ids)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3): model(ids)
        torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total / max(e.count, 1)) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1] * r[2])
tot = sum(r[1] * r[2] for r in rows)
for k, c, t in rows[:12]:
    print("%-100s calls %5d avg %9.1f us  %5.1f%%" % (k[:100], c, t, 100 * c * t / tot))
print("device time per forward (%d layers): %.2f ms" % (cfg.num_hidden_layers, tot / 3 / 1e3))
