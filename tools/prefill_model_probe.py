"""BASELINE config 3: whole-model prefill, 8 x 2048 tokens, LLaMA-7B OneBit, module path."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
dev = torch.device("cuda:0")
cfg = OneBitLlamaConfig.llama_7b()
model = build_synthetic_model(cfg, seed=1, device=dev)
for impl, fused in (("eager", False), ("sdpa", False), ("sdpa", True)):
  model.set_attention(impl).set_fused_glue(fused)
  for (B, S) in ((1, 2048), (8, 2048)):
      ids = torch.randint(0, cfg.vocab_size, (B, S), generator=torch.Generator().manual_seed(0)).to(dev)
      with torch.no_grad():
          model(ids[:1, :128]); model(ids)
          torch.cuda.synchronize(); t0 = time.perf_counter()
          n = 2
          for _ in range(n): lg = model(ids)
          torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
      flops = 2.0 * B * S * 6476005376
      print(impl, "fused-glue" if fused else "torch-glue", "prefill B=%d S=%d: %.1f ms  %.0f tok/s  1-bit layers at %.0f TFLOP/s equivalent (whole forward incl. attention, lm_head)"
            % (B, S, dt * 1e3, B * S / dt, flops / dt / 1e12))
      del lg
