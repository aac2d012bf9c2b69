#!/usr/bin/env python3
"""model_rope.npz: the reference model (BitLlamaForCausalLMInf, build container only) with its two RoPE-scaling variants
(modeling_bitllama.py:123-165, 298-326): "linear" (positions / factor) and "dynamic" NTK (base rescaled once the sequence
exceeds max_position_embeddings -- max_position_embeddings is 16 here and the prompt has 24 tokens, so the rescaling
happens, and two cached decode steps follow).  Weights regenerate from synthetic_state_dict(cfg, seed=7); the fixture
holds ids and fp32 logits.  Usage: PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/transformers/src:. python tests/golden/gen_goldens_rope.py"""
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
    sys.exit("reference not present")
sys.path.insert(0, REF_SRC)
from transformers import BitLlamaConfig, BitLlamaForCausalLMInf  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from onebit_amd.llama import OneBitLlamaConfig, synthetic_state_dict  # noqa: E402

KW = dict(vocab_size=96, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=16)
out = {"cfg_" + k: np.array(v) for k, v in KW.items()}
g = torch.Generator().manual_seed(31)
ids = torch.randint(0, KW["vocab_size"], (1, 24), generator=g)
out["input_ids"] = ids.numpy()
sd = {k: (v if v.dtype == torch.int8 else v.float()) for k, v in synthetic_state_dict(OneBitLlamaConfig(**KW), seed=7, dtype=torch.float16).items()}
for name, rs in (("linear", {"type": "linear", "factor": 2.0}), ("dynamic", {"type": "dynamic", "factor": 2.0}), ("none", None)):
    model = BitLlamaForCausalLMInf(BitLlamaConfig(rope_scaling=rs, **KW))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model = model.float().eval()
    with torch.no_grad():
        o = model(ids, use_cache=True)
        lg = [o.logits.numpy()]
        tok = o.logits[:, -1].argmax(-1, keepdim=True)
        toks = [tok]
        past = o.past_key_values
        for _ in range(2):
            o = model(tok, past_key_values=past, use_cache=True)
            past = o.past_key_values
            lg.append(o.logits.numpy())
            tok = o.logits[:, -1].argmax(-1, keepdim=True)
            toks.append(tok)
    out["logits_" + name] = np.concatenate(lg, axis=1)            # [1, 26, vocab]
    out["greedy_" + name] = torch.cat(toks, dim=1).numpy()
assert np.abs(out["logits_linear"] - out["logits_none"]).max() > 1e-3 and np.abs(out["logits_dynamic"] - out["logits_none"]).max() > 1e-3
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_rope.npz"), **out)
print("wrote model_rope.npz", {k: v.shape for k, v in out.items()})
