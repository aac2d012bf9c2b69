#!/usr/bin/env python3
"""Full-WIDTH model goldens from the REAL reference model (build container only): config "c" = LLaMA-7B layer
widths (hidden 4096, intermediate 11008, 32 heads of 128) with 2 layers and a 512-token vocabulary, so that the
reference's ``BitLlamaForCausalLMInf`` (modeling_bitllama.py:1512-1611) runs on the CPU here in minutes.

Weights are NOT stored: they are ``synthetic_state_dict(cfg, seed=7, device="cpu")`` (bit-reproducible from the
seed, 52 MB packed); the fixture holds only token ids and the reference's logits / greedy tokens:

  short   prompt of 12 tokens: prefill logits + 4 incremental decode steps (KV cache), fp16 and fp32 parameters
  batch   32 sequences x 6 prompt tokens in one batched reference call, last-position logits + 3 batched decode
          steps, fp16 and fp32 (pins the 32-slot native batched step and the batcher at full width)
  long    one 4096-token prompt (every projection of this width takes the LDS-DMA prefill GEMM from T = 4096 in the build), fp16 and fp32: logits of
          16 positions (the last 8 and 8 seeded ones)

Config "d" (round 4) = the same widths at FULL DEPTH: 32 layers (the whole LLaMA-7B decoder stack,
modeling_bitllama.py:1287-1319 layer loop; vocabulary 512), seed 11 — pins error growth through 32 x 7
LayerNorm-terminated projections:

  short   12-token prompt: prefill logits + 2 incremental decode steps, fp16 and fp32
  batch   8 sequences x 6 prompt tokens in one batched reference call + 2 batched decode steps, fp16 and fp32

Config "e" (round 4) = LLaMA-13B layer widths (hidden 5120, intermediate 13824, 40 heads of 128; BASELINE configs 4 and 5), 2 layers,
seed 13: 12-token prompt + 3 decode steps, 8 sequences batched + 2 batched steps, fp16 and fp32 -- pins the 13B launch geometries
(one projection per workgroup with 80 / 124 workgroups per projection) and the K-sharded module path to the reference.

Output: tests/golden/model_wide_c.npz / model_wide_d.npz / model_wide_e.npz (data only).  Usage (from the repo root):
  PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/transformers/src:. python tests/golden/gen_goldens_model_wide.py [c|d|e] [parts...]
"""
import importlib.metadata as md
import os
import sys
import time

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
    "c": dict(kw=dict(vocab_size=512, hidden_size=4096, intermediate_size=11008, num_hidden_layers=2,
                      num_attention_heads=32, max_position_embeddings=4352),
              seed=7, parts=["short", "batch", "long"], steps=4, batch=(32, 6), batch_steps=3),
    "d": dict(kw=dict(vocab_size=512, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                      num_attention_heads=32, max_position_embeddings=256),
              seed=11, parts=["short", "batch"], steps=2, batch=(8, 6), batch_steps=2),
    "e": dict(kw=dict(vocab_size=512, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2,
                      num_attention_heads=40, max_position_embeddings=256),
              seed=13, parts=["short", "batch"], steps=3, batch=(8, 6), batch_steps=2),
}
LONG = 4096


def main(name, parts):
    C = CONFIGS[name]
    KW, SEED = C["kw"], C["seed"]
    parts = parts or C["parts"]
    cfg = OneBitLlamaConfig(**KW)
    path = f"{OUT}/model_wide_{name}.npz"
    out = dict(np.load(path)) if os.path.exists(path) else {}
    out.update({"cfg_" + k: np.array(v) for k, v in KW.items()})
    out["seed"] = np.array(SEED)
    sd16 = synthetic_state_dict(cfg, seed=SEED, dtype=torch.float16)
    g = torch.Generator().manual_seed(2024)
    ids = torch.randint(0, cfg.vocab_size, (1, 12), generator=g)
    bids = torch.randint(0, cfg.vocab_size, C["batch"], generator=g)
    lids = torch.randint(0, cfg.vocab_size, (1, LONG), generator=g)
    lpos = np.array(sorted(set(range(LONG - 8, LONG)) | set(np.random.default_rng(5).integers(0, LONG - 8, 8).tolist())))
    out["input_ids"], out["batch_ids"] = ids.numpy(), bids.numpy()
    if "long" in C["parts"]:
        out["long_ids"], out["long_pos"] = lids.numpy(), lpos
    for dt, dn in ((torch.float16, "f16"), (torch.float32, "f32")):
        model = BitLlamaForCausalLMInf(BitLlamaConfig(**KW))
        sd = {k: (v if v.dtype == torch.int8 else v.to(dt)) for k, v in sd16.items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
        model = model.to(dt).eval()
        with torch.no_grad():
            if "short" in parts:
                t0 = time.time()
                o = model(ids, use_cache=True)
                out[f"prefill_logits_{dn}"] = o.logits.float().numpy()
                past = o.past_key_values
                tok = o.logits[:, -1].argmax(-1, keepdim=True)
                toks, step_logits = [tok], []
                for _ in range(C["steps"]):
                    o = model(tok, past_key_values=past, use_cache=True)
                    past = o.past_key_values
                    step_logits.append(o.logits.float().numpy())
                    tok = o.logits[:, -1].argmax(-1, keepdim=True)
                    toks.append(tok)
                out[f"decode_logits_{dn}"] = np.concatenate(step_logits, axis=1)
                out[f"greedy_{dn}"] = torch.cat(toks, dim=1).numpy()
                print(dn, "short", round(time.time() - t0, 1), "s greedy", out[f"greedy_{dn}"].tolist(), flush=True)
            if "batch" in parts:
                t0 = time.time()
                o = model(bids, use_cache=True)
                past = o.past_key_values
                lg = [o.logits[:, -1:].float().numpy()]
                tok = o.logits[:, -1].argmax(-1, keepdim=True)
                btoks = [tok]
                for _ in range(C["batch_steps"]):
                    o = model(tok, past_key_values=past, use_cache=True)
                    past = o.past_key_values
                    lg.append(o.logits.float().numpy())
                    tok = o.logits[:, -1].argmax(-1, keepdim=True)
                    btoks.append(tok)
                out[f"batch_logits_{dn}"] = np.concatenate(lg, axis=1)            # [32, 4, vocab]: last prompt position + 3 steps
                out[f"batch_greedy_{dn}"] = torch.cat(btoks, dim=1).numpy()      # [32, 4]
                print(dn, "batch", round(time.time() - t0, 1), "s", flush=True)
            if "long" in parts:
                t0 = time.time()
                o = model(lids, use_cache=False)
                out[f"long_logits_{dn}"] = o.logits[0, lpos].float().numpy()      # [16, vocab]
                print(dn, "long", round(time.time() - t0, 1), "s", flush=True)
        del model
        np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; keys", sorted(out))


if __name__ == "__main__":
    torch.set_num_threads(8)
    args = sys.argv[1:]
    name = args.pop(0) if args and args[0] in CONFIGS else "c"
    main(name, args)
