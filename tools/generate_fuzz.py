#!/usr/bin/env python3
"""generate_native against generate on random prompt lengths / generation lengths that cross the decode engine's regime boundaries (64 / 160
cached tokens, key-block split counts, prefill chunks), B = 1 and B > 1; tokens equal up to near-ties of the module path's logits.
python tools/generate_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = OneBitLlamaConfig(vocab_size=256, hidden_size=1024, intermediate_size=2816, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=4,
                        max_position_embeddings=1024)
model = build_synthetic_model(cfg, seed=77, device=dev)
g = torch.Generator().manual_seed(5)
bad = ties = 0
for c in range(cases):
    B = 1 if c % 3 else int(torch.randint(2, 6, (1,), generator=g))
    S = int(torch.randint(1, 500 if B == 1 else 120, (1,), generator=g))
    n = int(torch.randint(2, 260 if B == 1 else 60, (1,), generator=g))
    ids = torch.randint(0, cfg.vocab_size, (B, S), generator=g).to(dev)
    ref = model.generate(ids, n)
    got = model.generate_native(ids, n)
    assert got.shape == ref.shape
    for r in range(B):
        a, b = got[r].tolist(), ref[r].tolist()
        if a != b:
            j = next(i for i in range(len(a)) if a[i] != b[i])
            lg = model(torch.tensor([b[:j]], device=dev))[0, -1].float()
            d = abs(float(lg[a[j]] - lg[b[j]])) / float(lg.abs().max())
            ties += 1
            if d > 2e-2:
                bad += 1
                print("case %d row %d: diverges at %d beyond a near-tie (%.4f)" % (c, r, j, d), flush=True)
    print("case %d: B %d, prompt %d, %d new tokens: ok" % (c, B, S, n), flush=True)
print("GENFUZZ %s: %d cases, %d near-tie divergences, %d bad" % ("FAILED" if bad else "ok", cases, ties, bad))
sys.exit(1 if bad else 0)
