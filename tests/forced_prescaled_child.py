"""Child process of test_prescaled_prefill_route_forced_on_a_small_model: OB_GEMM3=2 (read once per process) makes
the LDS-DMA GEMM -- and with it the pre-scaled prefill route of the fused forward -- eligible on a small model.
Compares the fused route's logits with the module path's (eager attention, per-op glue)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1536, num_hidden_layers=3,
                            num_attention_heads=8, num_key_value_heads=4, max_position_embeddings=256)
    model = build_synthetic_model(cfg, seed=8, device=dev)
    ids = torch.randint(0, 512, (2, 128), generator=torch.Generator().manual_seed(3)).to(dev)     # T = 256 rows
    ref = model(ids)
    att = model.model.layers[0].self_attn
    assert att.q_proj.prescaled_ok(256) and model.model.layers[0].mlp.down_proj.prescaled_ok(256), "route not forced"
    got = model.set_fused_glue(True).set_attention("sdpa")(ids)
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= 6e-3 * scale, (err, scale)
    print("forced-prescaled ok %.3g of %.3g" % (err, scale))


if __name__ == "__main__":
    main()
