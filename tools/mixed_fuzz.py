#!/usr/bin/env python3
"""Randomised check of onebit_mixed_step: random mixes of decoding slots and prompt chunks (row counts crossing every route threshold: 64 / 65 /
128 / 192 / 320 / 321 ...), every item's greedy token against the module path run on the item's whole history (near-ties of the logits
tolerated and re-synchronised).  python tools/mixed_fuzz.py [seeds] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import MixedStep
dev = torch.device("cuda:0")
seeds, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 24
CFGS = [dict(vocab_size=512, hidden_size=1024, intermediate_size=2816, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, max_position_embeddings=1024),
        dict(vocab_size=512, hidden_size=2048, intermediate_size=5632, num_hidden_layers=2, num_attention_heads=16, max_position_embeddings=1024),
        dict(vocab_size=384, hidden_size=512, intermediate_size=1536, num_hidden_layers=3, num_attention_heads=8, max_position_embeddings=1024, attention_bias=True),
        dict(vocab_size=256, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2, num_attention_heads=40, max_position_embeddings=1024),     # 13B widths
        dict(vocab_size=256, hidden_size=4096, intermediate_size=11008, num_hidden_layers=2, num_attention_heads=32, max_position_embeddings=1024)]     # 7B widths
bad = ties = checked = 0
for seed in range(seeds):
    cfg = OneBitLlamaConfig(**CFGS[seed % len(CFGS)])
    model = build_synthetic_model(cfg, seed=100 + seed, device=dev)
    slots, max_len = 8, 900
    shape = (slots, cfg.num_key_value_heads, max_len, cfg.head_dim)
    caches = [(torch.zeros(shape, dtype=torch.float16, device=dev), torch.zeros(shape, dtype=torch.float16, device=dev)) for _ in range(cfg.num_hidden_layers)]
    ms = MixedStep(model, caches, slots, max_len, max_rows=64, keep_logits=True)
    g = torch.Generator().manual_seed(seed)
    hist = {s: [] for s in range(slots)}          # tokens in the cache per slot
    pend = {s: None for s in range(slots)}        # the token a decoding slot feeds next
    for step in range(steps):
        target = [3, 40, 64, 65, 100, 128, 129, 190, 192, 200, 256, 300, 320, 321, 350, 400, 520, 650, 700][int(torch.randint(0, 19, (1,), generator=g))]
        items, rows = [], 0
        order = torch.randperm(slots, generator=g).tolist()
        for s in order:
            if len(hist[s]) + 2 >= max_len - 1:
                hist[s], pend[s] = [], None               # recycle the slot
            if pend[s] is not None and torch.rand(1, generator=g) < 0.7:
                items.append((s, len(hist[s]), [pend[s]])); rows += 1
            elif rows < target:
                room = max_len - 2 - len(hist[s])
                n = min(room, max(1, min(target - rows, int(torch.randint(1, 400, (1,), generator=g)))))
                if n < 1:
                    continue
                toks = torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist()
                if pend[s] is not None:
                    toks[0] = pend[s]
                items.append((s, len(hist[s]), toks)); rows += n
        if not items:
            continue
        nxt = ms.launch(items).clone()
        torch.cuda.synchronize()
        lg = ms.logits[:len(items)].float()
        for i, (s, start, toks) in enumerate(items):
            hist[s] = hist[s] + toks
            assert start + len(toks) == len(hist[s])
            ref = model(torch.tensor([hist[s]], device=dev))[0, -1].float()
            scale = float(ref.abs().max())
            err = float((lg[i] - ref).abs().max())
            checked += 1
            if err > 1.5e-2 * scale:
                bad += 1
                print("seed %d step %d rows %d item %d (slot %d, start %d, %d tokens): logits differ by %.4f of scale %.3f" % (seed, step, rows, i, s, start, len(toks), err, scale), flush=True)
            t, r = int(nxt[i]), int(ref.argmax())
            if t != r:
                ties += 1
                if abs(float(ref[t] - ref[r])) > 2e-2 * scale:
                    bad += 1
                    print("seed %d step %d item %d: token %d vs %d beyond a near-tie" % (seed, step, i, t, r), flush=True)
            pend[s] = r                                   # continue on the reference's token
    print("seed %d (%s): ok so far, %d items checked, %d near-ties, %d bad" % (seed, "H=%d" % cfg.hidden_size, checked, ties, bad), flush=True)
print("FUZZ %s: %d items, %d near-ties, %d bad" % ("FAILED" if bad else "ok", checked, ties, bad))
sys.exit(1 if bad else 0)
