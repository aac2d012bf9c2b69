"""Checkpoint format: key layout identical to the reference's inference state dict (recorded in
tests/golden/model_tiny_*.npz from BitLlamaForCausalLMInf), save/load round trip, converter key logic."""
import os

import numpy as np
import pytest
import torch

from onebit_amd import checkpoint as C
from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM, synthetic_state_dict


def test_state_dict_keys_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "model_tiny_b.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    ref_keys = sorted(k[3:] for k in z.files if k.startswith("sd_"))
    assert sorted(model.state_dict().keys()) == ref_keys
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == z["sd_" + k].shape, k
        assert (v.dtype == torch.int8) == (z["sd_" + k].dtype == np.int8), k


def test_save_load_roundtrip(tmp_path):
    cfg = OneBitLlamaConfig(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                            num_attention_heads=2, max_position_embeddings=32)
    model = OneBitLlamaForCausalLM(cfg, torch.float16)
    model.load_state_dict(synthetic_state_dict(cfg, seed=3))
    C.save_inference_checkpoint(model, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["config.json", "pytorch_model.bin"]
    back = C.load_inference_checkpoint(str(tmp_path), device="cpu")
    assert back.config == cfg
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), back.state_dict().items()):
        assert k1 == k2 and v1.dtype == v2.dtype and torch.equal(v1, v2)
    # fp32 checkpoint on disk (the released ones are, checkpoints/README.md:10): floats are cast, int8 is not
    sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    torch.save(sd32, os.path.join(tmp_path, "pytorch_model.bin"))
    back = C.load_inference_checkpoint(str(tmp_path), device="cpu")
    assert back.model.layers[0].mlp.up_proj.weight.dtype == torch.int8
    assert back.model.layers[0].mlp.up_proj.weight_scale.dtype == torch.float16


def test_rejects_foreign_or_unconverted(tmp_path):
    import json
    with open(tmp_path / "config.json", "w") as f:
        json.dump({"model_type": "llama"}, f)
    with pytest.raises(ValueError):
        C.config_from_json(str(tmp_path / "config.json"))
    cfg = OneBitLlamaConfig(vocab_size=16, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1)
    with open(tmp_path / "config.json", "w") as f:
        json.dump(C.config_to_json(cfg), f)
    torch.save({"model.layers.0.mlp.up_proj.weight": torch.randn(64, 32)}, tmp_path / "pytorch_model.bin")
    with pytest.raises(ValueError, match="latent"):
        C.load_inference_checkpoint(str(tmp_path), device="cpu")


def test_converter_selects_only_bitlinear_weights(monkeypatch):
    import onebit_amd.bitnet as B
    calls = []

    def fake_pack(w):
        calls.append(tuple(w.shape))
        return torch.zeros(w.shape[0], w.shape[1] // 8, dtype=torch.int8)

    monkeypatch.setattr(B, "pack_signs", fake_pack)
    sd = {"model.embed_tokens.weight": torch.randn(16, 32), "lm_head.weight": torch.randn(16, 32),
          "model.layers.0.mlp.up_proj.weight": torch.randn(64, 32), "model.layers.0.mlp.up_proj.weight_scale": torch.ones(64),
          "model.layers.0.mlp.up_proj.input_factor": torch.ones(32), "model.norm.weight": torch.ones(32)}
    out = C.convert_train_state_dict(sd, device="cpu")
    assert calls == [(64, 32)]
    assert out["model.layers.0.mlp.up_proj.weight"].dtype == torch.int8
    assert out["lm_head.weight"].dtype == torch.float32 and out["model.embed_tokens.weight"].shape == (16, 32)
