"""Whole-model prefill (8 x 2048, 7B, sdpa attention + fused glue) for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
dev = torch.device("cuda:0")
cfg = OneBitLlamaConfig.llama_7b()
model = build_synthetic_model(cfg, seed=1, device=dev)
model.set_attention("sdpa").set_fused_glue(True)
ids = torch.randint(0, cfg.vocab_size, (8, 2048), generator=torch.Generator().manual_seed(0)).to(dev)
with torch.no_grad():
    for _ in range(3):
        lg = model(ids)
torch.cuda.synchronize()
