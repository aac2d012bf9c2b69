"""Inference-checkpoint I/O in the reference's on-disk format -- SURVEY.md section 8(f) rank 2.

The reference converter (``scripts/convert_llama_to_infer_ckpt.py``) loads a KD-trained
``BitLlamaForCausalLM`` (latent fp weights), packs ``sign(W)`` of every ``BitLinear`` with
``fp16_to_int8`` (:26-34) and ``save_pretrained``s the result: a directory with ``config.json``
(``model_type: "bitllama"``, ``configuration_bitllama.py:112``) and ``pytorch_model.bin`` (torch
pickle, keys ``model.layers.{i}.{self_attn.{q,k,v,o}_proj|mlp.{gate,up,down}_proj}.{weight: int8
[N, K/8], weight_scale: [N], input_factor: [K]}`` + embeddings, RMSNorm weights, ``lm_head``).
This module reads and writes exactly that layout and restates the conversion step.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import torch

from .llama import OneBitLlamaConfig, OneBitLlamaForCausalLM

_CFG_KEYS = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
             "num_key_value_heads", "max_position_embeddings", "rms_norm_eps", "rope_theta", "attention_bias", "rope_scaling")
WEIGHTS_NAME = "pytorch_model.bin"


def config_from_json(path: str) -> OneBitLlamaConfig:
    with open(path) as f:
        raw = json.load(f)
    if raw.get("model_type", "bitllama") != "bitllama":
        raise ValueError(f"not a OneBit checkpoint: model_type={raw.get('model_type')!r}")
    if raw.get("pretraining_tp", 1) != 1:
        raise NotImplementedError("pretraining_tp > 1 (dead code in the reference's Inf classes, SURVEY.md fact 9)")
    if raw.get("hidden_act", "silu") != "silu":
        raise NotImplementedError("only hidden_act='silu'")
    return OneBitLlamaConfig(**{k: raw[k] for k in _CFG_KEYS if k in raw and raw[k] is not None})


def config_to_json(cfg: OneBitLlamaConfig) -> dict:
    d = {k: getattr(cfg, k) for k in _CFG_KEYS}
    d.update(model_type="bitllama", architectures=["BitLlamaForCausalLMInf"], hidden_act="silu",
             pretraining_tp=1, tie_word_embeddings=False, torch_dtype="float16")
    return d


def is_packed_state_dict(sd: Dict[str, torch.Tensor]) -> bool:
    return any(k.endswith("_proj.weight") and v.dtype in (torch.int8, torch.uint8) for k, v in sd.items())


def convert_train_state_dict(sd: Dict[str, torch.Tensor], device="cuda") -> Dict[str, torch.Tensor]:
    """Training checkpoint (latent fp ``BitLinear.weight`` [N, K]) -> inference state dict: every
    ``X.weight`` that has ``X.weight_scale`` and ``X.input_factor`` siblings is replaced by
    ``fp16_to_int8(sign(weight))`` (convert_llama_to_infer_ckpt.py:29-34); everything else is copied."""
    from .bitnet import pack_signs
    out = {}
    for k, v in sd.items():
        stem = k[:-len(".weight")] if k.endswith(".weight") else None
        if stem and stem + ".weight_scale" in sd and stem + ".input_factor" in sd and v.is_floating_point() and v.dim() == 2:
            out[k] = pack_signs(v.to(device)).cpu()
        else:
            out[k] = v
    return out


def load_inference_checkpoint(path: str, device="cuda", dtype=torch.float16) -> OneBitLlamaForCausalLM:
    """``BitLlamaForCausalLMInf.from_pretrained(path, torch_dtype=dtype)`` for this build's model: int8
    packed weights stay int8, floating tensors are cast to ``dtype`` (modeling_utils.py:696)."""
    cfg = config_from_json(os.path.join(path, "config.json"))
    sd = torch.load(os.path.join(path, WEIGHTS_NAME), map_location="cpu", weights_only=True)
    if not is_packed_state_dict(sd):
        raise ValueError("checkpoint holds latent weights; run convert_train_state_dict first")
    sd = {k: (v if not v.is_floating_point() else v.to(dtype)) for k, v in sd.items() if "rotary_emb.inv_freq" not in k}
    model = OneBitLlamaForCausalLM(cfg, dtype)
    model.load_state_dict(sd)
    return model.to(device).eval()


def save_inference_checkpoint(model: OneBitLlamaForCausalLM, path: str) -> Tuple[str, str]:
    os.makedirs(path, exist_ok=True)
    cfg_path, w_path = os.path.join(path, "config.json"), os.path.join(path, WEIGHTS_NAME)
    with open(cfg_path, "w") as f:
        json.dump(config_to_json(model.config), f, indent=2)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, w_path)
    return cfg_path, w_path
