#!/usr/bin/env python3
"""Model-level golden fixtures from the REAL reference model (build container only).

Imports the reference's forked transformers (read-only, PYTHONDONTWRITEBYTECODE=1) with the
two over-strict dependency pins shimmed, builds a tiny
``BitLlamaForCausalLMInf``, assigns a seeded synthetic inference checkpoint to it (the same
``synthetic_state_dict`` the build uses), and records: the state dict, prompt ids, prefill
logits, the logits of 4 incremental decode steps (KV cache) and the greedy tokens, for fp32
and fp16 parameters.  Output: tests/golden/model_tiny_{a,b}.npz (data only).

Usage (from the repo root):
  PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/transformers/src:. \
      python tests/golden/gen_goldens_model.py
"""
import importlib.metadata as md
import os
import sys

import numpy as np
import torch

_orig = md.version
_fake = {"tokenizers": "0.14.1", "huggingface-hub": "0.17.3", "huggingface_hub": "0.17.3"}
md.version = lambda n: _fake.get(n, _orig(n))

REF_SRC = "/root/reference/transformers/src"
if not os.path.isdir(REF_SRC):
    sys.exit("reference not present; model fixtures can only be regenerated in the build container")
sys.path.insert(0, REF_SRC)
from transformers import BitLlamaConfig, BitLlamaForCausalLMInf  # noqa: E402  (the reference fork)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from onebit_amd.llama import OneBitLlamaConfig, synthetic_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CONFIGS = {
    # a: the survey's tiny shape; intermediate 688 is not a multiple of 32 (generic kernel for down_proj)
    "a": dict(vocab_size=128, hidden_size=256, intermediate_size=688, num_hidden_layers=2,
              num_attention_heads=4, max_position_embeddings=64),
    # b: every projection on the MFMA path; head_dim 64
    "b": dict(vocab_size=96, hidden_size=128, intermediate_size=352, num_hidden_layers=2,
              num_attention_heads=2, max_position_embeddings=64),
    # bias: config.attention_bias = True -- q / k / v / o_proj carry a bias added after their LayerNorm
    # (modeling_bitllama.py:451-454, bitnet.py:119-120); shapes of "b", 3 layers
    "bias": dict(vocab_size=96, hidden_size=128, intermediate_size=352, num_hidden_layers=3,
                 num_attention_heads=2, max_position_embeddings=64, attention_bias=True),
}


def run(name, kw):
    cfg = OneBitLlamaConfig(**kw)
    out = {"cfg_" + k: np.array(v) for k, v in kw.items()}
    sd16 = synthetic_state_dict(cfg, seed=7, dtype=torch.float16)
    for k, v in sd16.items():
        out["sd_" + k] = v.numpy()
    g = torch.Generator().manual_seed(99)
    ids = torch.randint(0, cfg.vocab_size, (1, 8), generator=g)
    out["input_ids"] = ids.numpy()
    for dt, dn in ((torch.float32, "f32"), (torch.float16, "f16")):
        rcfg = BitLlamaConfig(**kw)
        model = BitLlamaForCausalLMInf(rcfg)
        sd = {k: (v if v.dtype == torch.int8 else v.to(dt)) for k, v in sd16.items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
        model = model.to(dt).eval()
        for n_, p_ in model.named_parameters():
            if "proj.weight" in n_ and not n_.endswith(("weight_scale",)):
                assert p_.dtype == torch.int8, n_
        with torch.no_grad():
            o = model(ids, use_cache=True)
            out[f"prefill_logits_{dn}"] = o.logits.numpy()
            past = o.past_key_values
            tok = o.logits[:, -1].argmax(-1, keepdim=True)
            toks, step_logits = [tok], []
            for _ in range(4):
                o = model(tok, past_key_values=past, use_cache=True)
                past = o.past_key_values
                step_logits.append(o.logits.numpy())
                tok = o.logits[:, -1].argmax(-1, keepdim=True)
                toks.append(tok)
            out[f"decode_logits_{dn}"] = np.concatenate(step_logits, axis=1)
            out[f"greedy_{dn}"] = torch.cat(toks, dim=1).numpy()
            gen = model.generate(ids, max_new_tokens=5, do_sample=False)
            assert torch.equal(gen[:, 8:], torch.cat(toks, dim=1)), "generate() disagrees with manual greedy loop"
            # logit margin of the chosen tokens, to judge how robust greedy parity is
            last = np.concatenate([out[f"prefill_logits_{dn}"][:, -1:], out[f"decode_logits_{dn}"]], axis=1)[0]
            srt = np.sort(last, axis=-1)
            out[f"margin_{dn}"] = srt[:, -1] - srt[:, -2]
    np.savez_compressed(f"{OUT}/model_tiny_{name}.npz", **out)
    print(name, "greedy f32", out["greedy_f32"], "f16", out["greedy_f16"],
          "margins f16", np.round(out["margin_f16"], 3), os.path.getsize(f"{OUT}/model_tiny_{name}.npz"))


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1:]                       # e.g. `gen_goldens_model.py bias`: regenerate one fixture
    for n, kw in CONFIGS.items():
        if not only or n in only:
            run(n, kw)
