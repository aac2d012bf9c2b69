"""Round 6: ragged token rows (several sequences in one call) and the native mixed prefill + decode step of continuous batching
(BASELINE config 5).  Every kernel entry point through the C ABI:
  * onebit_rows_qkv_rope_ragged   bit-identical to onebit_rows_qkv_rope on the same rows
  * onebit_attention_ragged       bit-identical to onebit_attention_prefill per segment; fp32 reference (modeling_bitllama.py:546-563)
  * onebit_attention_decode_rows  fp32 reference across the split boundaries, GQA, head_dim 64, idle rows, deterministic combine
  * onebit_mixed_step             tokens equal single-sequence generate; logits of the rows against the module path
"""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _lib():
    from onebit_amd import _lib as L
    return L, L.load()


def _sp():
    return torch.cuda.current_stream(torch.device(DEV)).cuda_stream


def _rope_tables(D, n, dev):
    from onebit_amd.llama import rope_tables
    return rope_tables(D, n, 10000.0, dev, torch.float16)


def _segs(lst):
    from onebit_amd.engine import _Seg
    return (_Seg * len(lst))(*[_Seg(*g) for g in lst])


@pytest.mark.parametrize("H,Hkv,D", [(4, 4, 64), (8, 2, 128), (32, 32, 128)])
def test_rope_ragged_is_the_uniform_kernel_on_the_same_rows(H, Hkv, D):
    L, lib = _lib()
    dev = torch.device(DEV)
    g = torch.Generator(device="cpu").manual_seed(H * 7 + D)
    B, S, past, max_len, slots = 3, 5, 7, 32, 6
    T, NQ, NK = B * S, H * D, Hkv * D
    uq, uk, uv = [(torch.randn(T, n, generator=g) * 3).half().to(dev) for n in (NQ, NK, NK)]
    cos, sin = _rope_tables(D, max_len, dev)
    # uniform: sequences 0..B-1 in slots 0..B-1
    q0 = torch.zeros(T, NQ, dtype=torch.float16, device=dev)
    k0 = torch.zeros(slots, Hkv, max_len, D, dtype=torch.float16, device=dev)
    v0 = torch.zeros_like(k0)
    L.check(lib.onebit_rows_qkv_rope(uq.data_ptr(), uk.data_ptr(), uv.data_ptr(), cos.data_ptr(), sin.data_ptr(), q0.data_ptr(), k0.data_ptr(),
                                     v0.data_ptr(), B, S, H, Hkv, D, past, max_len, max_len, 1e-5, L.FLAG_Q_TOKEN_MAJOR, _sp()), "rope")
    # ragged: the same rows, each told its (slot, position); slots permuted, one extra idle row
    perm = [4, 0, 2]
    row_slot = torch.tensor([perm[t // S] for t in range(T)] + [99], dtype=torch.int32, device=dev)
    row_pos = torch.tensor([past + t % S for t in range(T)] + [-1], dtype=torch.int32, device=dev)
    pad = lambda u: torch.cat([u, torch.full_like(u[:1], 7.0)])
    q1 = torch.zeros(T + 1, NQ, dtype=torch.float16, device=dev)
    k1 = torch.zeros_like(k0)
    v1 = torch.zeros_like(k0)
    uq1, uk1, uv1 = pad(uq), pad(uk), pad(uv)
    L.check(lib.onebit_rows_qkv_rope_ragged(uq1.data_ptr(), uk1.data_ptr(), uv1.data_ptr(), cos.data_ptr(), sin.data_ptr(), row_slot.data_ptr(),
                                            row_pos.data_ptr(), q1.data_ptr(), k1.data_ptr(), v1.data_ptr(), None, None, None, T + 1, H, Hkv, D, slots,
                                            max_len, max_len, 1e-5, _sp()), "rope ragged")
    torch.cuda.synchronize()
    assert torch.equal(q1[:T], q0)
    assert float(q1[T].abs().max()) == 0.0                                  # the idle row wrote nothing
    for b in range(B):
        assert torch.equal(k1[perm[b]], k0[b]) and torch.equal(v1[perm[b]], v0[b])
    untouched = [s for s in range(slots) if s not in perm]
    assert float(k1[untouched].abs().max()) == 0.0 and float(v1[untouched].abs().max()) == 0.0


def _ref_attention(q, k, v, past):
    """q [n, H, D]; k, v [Hkv, L, D] (L = past + n): causal attention in fp32."""
    n, H, D = q.shape
    Hkv = k.shape[0]
    kk = k.float().repeat_interleave(H // Hkv, dim=0)
    vv = v.float().repeat_interleave(H // Hkv, dim=0)
    s = torch.einsum("nhd,hld->hnl", q.float(), kk) / math.sqrt(D)
    Lk = kk.shape[1]
    mask = torch.arange(Lk, device=q.device)[None, :] > (past + torch.arange(n, device=q.device))[:, None]
    s = s.masked_fill(mask[None], float("-inf"))
    return torch.einsum("hnl,hld->nhd", torch.softmax(s, -1), vv)


@pytest.mark.parametrize("H,Hkv,D", [(4, 4, 64), (8, 2, 128)])
def test_attention_ragged_equals_per_segment_prefill_attention(H, Hkv, D):
    L, lib = _lib()
    dev = torch.device(DEV)
    g = torch.Generator(device="cpu").manual_seed(11 + D)
    max_len, slots = 700, 5
    # (n, slot, past): chunks shorter / longer than a 128-query block, with and without past, one single-token segment
    spec = [(130, 3, 0), (1, 0, 57), (64, 1, 200), (300, 4, 399), (17, 2, 5)]
    T = sum(n for n, _, _ in spec)
    q = torch.randn(T, H, D, generator=g).half().to(dev)
    kc = torch.randn(slots, Hkv, max_len, D, generator=g).half().to(dev)
    vc = torch.randn(slots, Hkv, max_len, D, generator=g).half().to(dev)
    hn = (0.5 + torch.rand(H * D, generator=g)).half().to(dev)
    segs, row = [], 0
    for n, slot, past in spec:
        segs.append((row, n, slot, past))
        row += n
    for h_next in (None, hn):
        o = torch.zeros(T, H, D, dtype=torch.float16, device=dev)
        L.check(lib.onebit_attention_ragged(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), None if h_next is None else h_next.data_ptr(),
                                            ctypes.cast(_segs(segs), ctypes.c_void_p), len(segs), H, Hkv, D, slots, max_len, _sp()), "ragged")
        torch.cuda.synchronize()
        for (row0, n, slot, past) in segs:
            o1 = torch.zeros(1, n, H, D, dtype=torch.float16, device=dev)
            qs = q[row0:row0 + n].reshape(1, n, H, D).contiguous()
            L.check(lib.onebit_attention_prefill(qs.data_ptr(), kc[slot:slot + 1].data_ptr(), vc[slot:slot + 1].data_ptr(), o1.data_ptr(),
                                                 None if h_next is None else h_next.data_ptr(), 1, n, H, Hkv, D, past, max_len, _sp()), "prefill")
            torch.cuda.synchronize()
            assert torch.equal(o[row0:row0 + n], o1[0]), (n, slot, past)
            if h_next is None:
                ref = _ref_attention(q[row0:row0 + n], kc[slot, :, :past + n], vc[slot, :, :past + n], past)
                assert float((o[row0:row0 + n].float() - ref).abs().max()) < 4e-3


@pytest.mark.parametrize("H,Hkv,D,chunk", [(4, 4, 64, 64), (8, 2, 128, 128), (40, 40, 128, 256)])
def test_attention_decode_rows_across_split_boundaries(H, Hkv, D, chunk):
    L, lib = _lib()
    dev = torch.device(DEV)
    g = torch.Generator(device="cpu").manual_seed(5 + chunk)
    max_len, slots = 1100, 12
    ctx = [1, 2, chunk - 1, chunk, chunk + 1, 2 * chunk, 2 * chunk + 1, 517, 1100, 3 * chunk + 7]       # keys attended per row
    rows = len(ctx) + 1                                                                                # + one idle row
    slot_of = [5, 0, 7, 3, 1, 9, 2, 11, 4, 8, 6]
    q = torch.randn(rows, H, D, generator=g).half().to(dev)
    kc = (torch.randn(slots, Hkv, max_len, D, generator=g) * 0.7).half().to(dev)
    vc = torch.randn(slots, Hkv, max_len, D, generator=g).half().to(dev)
    row_slot = torch.tensor(slot_of, dtype=torch.int32, device=dev)
    row_pos = torch.tensor([c - 1 for c in ctx] + [-1], dtype=torch.int32, device=dev)
    nsplit = -(-max_len // chunk)
    nb = int(lib.onebit_attention_decode_scratch_bytes(rows, H, nsplit))
    scratch = torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)
    outs = []
    for rep in range(3):
        o = torch.full((rows, H * D), 9.0, dtype=torch.float16, device=dev)
        L.check(lib.onebit_attention_decode_rows(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), None, row_slot.data_ptr(), row_pos.data_ptr(),
                                                 rows, H, Hkv, D, slots, max_len, chunk, nsplit, scratch.data_ptr(), scratch.numel(), _sp()), "fdec")
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])        # fixed-order combine, tickets back at zero
    o = outs[0].view(rows, H, D)
    assert float((o[-1] - 9.0).abs().max()) == 0.0                                # idle row untouched
    for r, c in enumerate(ctx):
        ref = _ref_attention(q[r:r + 1], kc[slot_of[r], :, :c], vc[slot_of[r], :, :c], c - 1)[0]
        err = float((o[r].float() - ref).abs().max())
        assert err < 4e-3, (r, c, err)
    # a split grid smaller than the cache (host-known context bound): the same results for the rows inside it
    short = [r for r, c in enumerate(ctx) if c <= 2 * chunk]
    o2 = torch.zeros(rows, H * D, dtype=torch.float16, device=dev)
    rp2 = row_pos.clone()
    rp2[[r for r in range(len(ctx)) if r not in short]] = -1
    torch.cuda.synchronize()
    assert int(scratch[:rows * H * 4].max()) == 0                                 # every ticket is back at zero
    L.check(lib.onebit_attention_decode_rows(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o2.data_ptr(), None, row_slot.data_ptr(), rp2.data_ptr(),
                                             rows, H, Hkv, D, slots, max_len, chunk, 2, scratch.data_ptr(), scratch.numel(), _sp()), "fdec")
    torch.cuda.synchronize()
    assert torch.equal(o2[short], outs[0][short])


def test_decode_rows_argument_errors():
    L, lib = _lib()
    dev = torch.device(DEV)
    t = torch.zeros(4096, dtype=torch.float16, device=dev)
    i = torch.zeros(16, dtype=torch.int32, device=dev)
    call = lambda **kw: lib.onebit_attention_decode_rows(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, i.data_ptr(), i.data_ptr(),
                                                        kw.get("rows", 1), 4, kw.get("hkv", 4), kw.get("D", 64), 1, 16, kw.get("chunk", 64),
                                                        kw.get("ns", 1), kw.get("scr", None), kw.get("nb", 0), _sp())
    assert call() == 0
    assert call(chunk=100) == -2 and call(D=12) == -2 and call(hkv=3) == -2
    assert call(ns=4) == -5                                                     # needs scratch
    assert call(rows=0) == 0
    torch.cuda.synchronize()


def _tiny(golden_dir, name, dev):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    return model.to(dev).eval()


def _near_tie(model, p, ref, got, dev):
    j = next(i for i in range(len(ref)) if got[i] != ref[i])
    lg = model(torch.tensor([p + ref[:j]], device=dev))[0, -1]
    return abs(float(lg[got[j]] - lg[ref[j]])) < 2e-2 * float(lg.abs().max())


@pytest.mark.parametrize("chunk", [None, 4])
def test_mixed_step_tokens_equal_single_sequence_generate(golden_dir, chunk):
    """Tiny golden model b through ContinuousBatcher: every step that carries prompt tokens runs onebit_mixed_step (asserted),
    with more requests than slots, whole prompts and chunked prompts next to decoding requests."""
    from onebit_amd.serving import ContinuousBatcher
    dev = torch.device(DEV)
    model = _tiny(golden_dir, "b", dev)
    V = model.config.vocab_size
    g = torch.Generator().manual_seed(5)
    reqs = [(torch.randint(0, V, (n,), generator=g).tolist(), m) for n, m in [(8, 6), (1, 9), (13, 3), (5, 1), (20, 7), (2, 12), (9, 5)]]
    for use_graph in (True, False):
        cb = ContinuousBatcher(model, max_batch=3, max_len=40, prefill_chunk=chunk, use_graph=use_graph)
        assert cb._mixed is not None
        rids = [cb.add_request(p, m) for p, m in reqs]
        out = cb.run()
        assert cb.mixed_steps > 0 and (use_graph or cb.mixed_steps == cb.steps)
        for rid, (p, m) in zip(rids, reqs):
            ref = model.generate(torch.tensor([p], device=dev), m)[0, len(p):].tolist()
            assert len(out[rid]) == m
            assert out[rid] == ref or _near_tie(model, p, ref, out[rid], dev), rid


def test_mixed_step_logits_against_the_module_path(golden_dir):
    """One mixed step (2 decode rows with history + 3 prompt chunks, one of them continuing a half-entered prompt) on the tiny golden
    model: fp16 logits of every item's last row against the module path run per sequence (eager attention, reference op order)."""
    from onebit_amd.engine import MixedStep
    dev = torch.device(DEV)
    model = _tiny(golden_dir, "b", dev)
    cfg = model.config
    V, slots, max_len = cfg.vocab_size, 6, 48
    g = torch.Generator().manual_seed(17)
    shape = (slots, cfg.num_key_value_heads, max_len, cfg.head_dim)
    caches = [(torch.zeros(shape, dtype=torch.float16, device=dev), torch.zeros(shape, dtype=torch.float16, device=dev)) for _ in range(cfg.num_hidden_layers)]
    ms = MixedStep(model, caches, slots, max_len, max_rows=8, keep_logits=True)        # max_rows 8: the step below makes it grow
    seqs = {s: torch.randint(0, V, (n,), generator=g).tolist() for s, n in [(0, 9), (1, 30), (2, 5), (4, 21), (5, 12)]}
    # history through the step itself: slots 0, 1 complete prompts minus the last token; slot 4 the first 8 tokens
    ms.launch([(0, 0, seqs[0][:8]), (1, 0, seqs[1][:29]), (4, 0, seqs[4][:8])])
    torch.cuda.synchronize()
    items = [(0, 8, seqs[0][8:9]), (2, 0, seqs[2]), (1, 29, seqs[1][29:30]), (4, 8, seqs[4][8:]), (5, 0, seqs[5])]
    nxt = ms.launch(items).clone()
    torch.cuda.synchronize()
    lg = ms.logits[:len(items)].float()
    for i, (slot, start, toks) in enumerate(items):
        ref = model(torch.tensor([seqs[slot]], device=dev))[0, -1]
        scale = float(ref.abs().max())
        assert float((lg[i] - ref).abs().max()) < 1e-2 * scale, (i, slot)
        assert int(nxt[i]) == int(lg[i].argmax())


def test_mixed_step_at_13b_widths_chunked_prompts_next_to_decoding_slots():
    """BASELINE config 5 widths (hidden 5120 / intermediate 13824 / 40 heads, 2 layers): 512-token prompts entering in chunks of 256
    next to decoding requests; tokens equal single-sequence generate (fp16 near-ties tolerated at the first divergence)."""
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    from onebit_amd.serving import ContinuousBatcher
    dev = torch.device(DEV)
    cfg = OneBitLlamaConfig(vocab_size=512, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2,
                            num_attention_heads=40, max_position_embeddings=640)
    model = build_synthetic_model(cfg, seed=13, device=dev)
    g = torch.Generator().manual_seed(3)
    reqs = [(torch.randint(0, 512, (n,), generator=g).tolist(), m) for n, m in [(12, 10), (512, 4), (3, 12), (512, 3), (40, 6), (300, 5)]]
    cb = ContinuousBatcher(model, max_batch=4, max_len=600, prefill_chunk=256, max_step_tokens=600)
    assert cb._mixed is not None and cb._native is not None
    rids = [cb.add_request(p, m) for p, m in reqs]
    out = cb.run()
    assert cb.mixed_steps >= 5 and cb.graph_steps > 0
    for rid, (p, m) in zip(rids, reqs):
        ref = model.generate(torch.tensor([p], device=dev), m)[0, len(p):].tolist()
        assert len(out[rid]) == m
        assert out[rid] == ref or _near_tie(model, p, ref, out[rid], dev), rid


@pytest.mark.parametrize("hidden,inter,heads,rows", [(5120, 13824, 40, 543), (5120, 13824, 40, 700), (4096, 11008, 32, 700),
                                                     (5120, 13824, 40, 95), (5120, 13824, 40, 200), (4096, 11008, 32, 131), (4096, 11008, 32, 65)])
def test_mixed_step_mid_size_routes(hidden, inter, heads, rows):
    """A few hundred rows at 13B / 7B widths (7B down_proj: K = 11008 splits into 5632 + 5376): q|k|v and gate|up take pre-scaled rows because the GROUP fills the chip (ob_gemm3_group_ok),
    o_proj / down_proj run as two K-slices whose fp32 sums the next row kernel adds (ob_gemm3_ksplit2); 65 .. 320 rows: every projection
    as balanced passes of <= 64 rows through the LDS-DMA skinny GEMM.  Logits of the prompt's last row and of the decode rows against
    the module path."""
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    from onebit_amd.engine import MixedStep
    dev = torch.device(DEV)
    cfg = OneBitLlamaConfig(vocab_size=512, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=2,
                            num_attention_heads=heads, max_position_embeddings=1024)
    model = build_synthetic_model(cfg, seed=29, device=dev)
    slots, max_len, n_dec = 4, 800, 3
    shape = (slots, cfg.num_key_value_heads, max_len, cfg.head_dim)
    caches = [(torch.zeros(shape, dtype=torch.float16, device=dev), torch.zeros(shape, dtype=torch.float16, device=dev)) for _ in range(cfg.num_hidden_layers)]
    ms = MixedStep(model, caches, slots, max_len, max_rows=1024, keep_logits=True)
    g = torch.Generator().manual_seed(rows)
    seqs = {s: torch.randint(0, 512, (n,), generator=g).tolist() for s, n in [(0, 20), (1, 7), (2, 33), (3, rows - n_dec)]}
    ms.launch([(s, 0, seqs[s][:-1]) for s in range(n_dec)])
    torch.cuda.synchronize()
    items = [(s, len(seqs[s]) - 1, seqs[s][-1:]) for s in range(n_dec)] + [(3, 0, seqs[3])]
    assert sum(len(t) for _, _, t in items) == rows
    nxt = ms.launch(items).clone()
    torch.cuda.synchronize()
    lg = ms.logits[:len(items)].float()
    for i, (slot, start, toks) in enumerate(items):
        ref = model(torch.tensor([seqs[slot]], device=dev))[0, -1].float()
        scale = float(ref.abs().max())
        assert float((lg[i] - ref).abs().max()) < 1e-2 * scale, (i, slot)
        assert int(nxt[i]) == int(lg[i].argmax())


@pytest.mark.parametrize("chunk,ctxs", [(64, [1, 5, 63, 64, 65, 130, 200]), (128, [7, 128, 129, 255, 256, 300, 2])])
def test_batched_step_keyblock_attention_matches_the_one_workgroup_form(chunk, ctxs):
    """onebit_decode_step_batched with attn_splits (LayerNorm + RoPE + append launch, then (head, slot, split) workgroups) against
    the one-workgroup-per-(head, slot) attention on the same caches: contexts on both sides of the split boundaries, an idle slot;
    logits within fp16 rounding of each other (the key-block form does not round the probabilities to fp16), equal greedy tokens."""
    from onebit_amd.engine import BatchedDecodeStep
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device(DEV)
    cfg = OneBitLlamaConfig(vocab_size=640, hidden_size=1024, intermediate_size=2816, num_hidden_layers=2,
                            num_attention_heads=8, max_position_embeddings=320)
    model = build_synthetic_model(cfg, seed=31, device=dev)
    B, max_len = len(ctxs) + 1, 320
    g = torch.Generator(device="cpu").manual_seed(chunk)
    base = [(torch.randn(B, cfg.num_key_value_heads, max_len, cfg.head_dim, generator=g) * 0.5).half().to(dev) for _ in range(2 * cfg.num_hidden_layers)]
    toks = torch.randint(0, cfg.vocab_size, (B,), generator=g).to(torch.int32)
    pos = torch.tensor([c - 1 for c in ctxs] + [-1], dtype=torch.int32)
    res = []
    for splits in (0, -(-max_len // chunk)):
        caches = [(base[2 * l].clone(), base[2 * l + 1].clone()) for l in range(cfg.num_hidden_layers)]
        st = BatchedDecodeStep(model, caches, B, max_len, keep_logits=True, attn_splits=splits, attn_chunk=chunk)
        st.tokens.copy_(toks)
        st.pos.copy_(pos)
        st.launch()
        torch.cuda.synchronize()
        res.append((st.logits.float().clone(), st.next_tokens.clone(), caches))
    (l0, t0, c0), (l1, t1, c1) = res
    live = slice(0, B - 1)
    scale = float(l0[live].abs().max())
    assert float((l0[live] - l1[live]).abs().max()) < 4e-3 * scale
    assert torch.equal(t0[live], t1[live])
    # the appended keys / values: the same arithmetic; the short form takes its LayerNorm statistics from the GEMM's tile partials,
    # the rope kernel reduces the rows itself -- mean / rstd differ in the last fp32 bit, the rows by an fp16 ulp here and there
    for (k0, v0), (k1, v1) in zip(c0, c1):
        assert float((k0.float() - k1.float()).abs().max()) <= 2.0 ** -9 * float(k0.abs().max())
        assert float((v0.float() - v1.float()).abs().max()) <= 2.0 ** -9 * float(v0.abs().max())
        assert float((k0 != k1).float().mean()) < 0.02


def test_engines_take_the_keyblock_graphs_at_long_contexts():
    """DecodeEngine (long_context_from = 8: one graph per power-of-two split count) and ContinuousBatcher (long_context_from = 8)
    on a small synthetic model: tokens equal the module path's generate."""
    from onebit_amd.engine import DecodeEngine
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    from onebit_amd.serving import ContinuousBatcher
    dev = torch.device(DEV)
    cfg = OneBitLlamaConfig(vocab_size=160, hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8,
                            max_position_embeddings=256)
    model = build_synthetic_model(cfg, seed=77, device=dev)
    V = model.config.vocab_size
    g = torch.Generator().manual_seed(23)
    p = torch.randint(0, V, (1, 5), generator=g).to(dev)
    ref = model.generate(p, 150)
    for la in ("keyblock", "pair"):
        eng = DecodeEngine(model, max_len=200, long_context_from=8, long_attention=la, attn_chunk=64)
        assert eng._keyblock == (la == "keyblock")
        out = eng.generate(p, 150)
        if not torch.equal(out, ref):
            j = int((out[0] != ref[0]).nonzero()[0])
            assert _near_tie(model, ref[0, :5].tolist(), ref[0, 5:].tolist(), out[0, 5:].tolist(), dev), (la, j)
        if la == "keyblock":
            assert sorted(eng._kb_graphs) == [1, 2, 4]
    reqs = [(torch.randint(0, V, (n,), generator=g).tolist(), m) for n, m in [(6, 90), (30, 70), (2, 110), (70, 30), (11, 60)]]
    cb = ContinuousBatcher(model, max_batch=3, max_len=160, long_context_from=8, attn_chunk=64)
    rids = [cb.add_request(pp, m) for pp, m in reqs]
    out = cb.run()
    assert cb.long_steps > 0 and len(cb._long) >= 2
    for rid, (pp, m) in zip(rids, reqs):
        r = model.generate(torch.tensor([pp], device=dev), m)[0, len(pp):].tolist()
        assert out[rid] == r or _near_tie(model, pp, r, out[rid], dev), rid


def _tile_partials(u):
    """Per-16-row-tile LayerNorm partials {sum, sum of squared deviations from the tile mean} of the rows of u [rows, n] in the layout
    the decode kernels publish them (ceil(n / 4096) * 512 floats per row, tile t at [2t], [2t + 1])."""
    rows, n = u.shape
    t = u.float().view(rows, n // 16, 16)
    sm = t.sum(-1)
    m2 = ((t - sm[..., None] / 16.0) ** 2).sum(-1)
    st = torch.zeros(rows, ((n + 4095) // 4096) * 512, dtype=torch.float32, device=u.device)
    st[:, 0:2 * (n // 16):2] = sm
    st[:, 1:2 * (n // 16):2] = m2
    return st.contiguous()


@pytest.mark.parametrize("H,Hkv,D,chunk,bias", [(8, 8, 64, 64, False), (8, 2, 128, 128, True), (32, 32, 128, 128, False)])
def test_fused_decode_rows_equals_rope_launch_plus_attention(H, Hkv, D, chunk, bias):
    """onebit_attention_decode_rows_fused (query formed inside the attention launch from the pre-LayerNorm rows + tile partials, new key /
    value from LDS, cache append by the last split's workgroup) against onebit_rows_qkv_rope_ragged + onebit_attention_decode_rows on the
    same inputs: outputs within fp16 rounding, the appended cache rows within an fp16 ulp (statistics from partials vs from the rows)."""
    L, lib = _lib()
    dev = torch.device(DEV)
    g = torch.Generator(device="cpu").manual_seed(3 * H + D)
    max_len, slots = 400, 7
    ctx = [1, chunk, chunk + 1, 2 * chunk + 5, 399, 77]
    rows = len(ctx) + 1
    NQ, NK = H * D, Hkv * D
    uq, uk, uv = [(torch.randn(rows, n, generator=g) * 2 + 0.3).half().to(dev) for n in (NQ, NK, NK)]
    bq, bk, bv = [(torch.randn(n, generator=g) * 0.3).half().to(dev) if bias else None for n in (NQ, NK, NK)]
    bp = lambda b: None if b is None else b.data_ptr()
    cos, sin = _rope_tables(D, max_len, dev)
    base_k = (torch.randn(slots, Hkv, max_len, D, generator=g) * 0.7).half().to(dev)
    base_v = torch.randn(slots, Hkv, max_len, D, generator=g).half().to(dev)
    row_slot = torch.tensor([3, 0, 6, 1, 5, 2, 4], dtype=torch.int32, device=dev)
    row_pos = torch.tensor([c - 1 for c in ctx] + [-1], dtype=torch.int32, device=dev)
    hn = (0.5 + torch.rand(NQ, generator=g)).half().to(dev)
    nsplit = -(-max_len // chunk)
    scratch = torch.zeros(max(int(lib.onebit_attention_decode_scratch_bytes(rows, H, nsplit)), 16), dtype=torch.uint8, device=dev)
    # reference route: rope / append launch, then the attention
    k0, v0 = base_k.clone(), base_v.clone()
    q = torch.zeros(rows, NQ, dtype=torch.float16, device=dev)
    o0 = torch.zeros(rows, NQ, dtype=torch.float16, device=dev)
    L.check(lib.onebit_rows_qkv_rope_ragged(uq.data_ptr(), uk.data_ptr(), uv.data_ptr(), cos.data_ptr(), sin.data_ptr(), row_slot.data_ptr(),
                                            row_pos.data_ptr(), q.data_ptr(), k0.data_ptr(), v0.data_ptr(), bp(bq), bp(bk), bp(bv), rows, H, Hkv, D,
                                            slots, max_len, max_len, 1e-5, _sp()), "rope")
    L.check(lib.onebit_attention_decode_rows(q.data_ptr(), k0.data_ptr(), v0.data_ptr(), o0.data_ptr(), hn.data_ptr(), row_slot.data_ptr(),
                                             row_pos.data_ptr(), rows, H, Hkv, D, slots, max_len, chunk, nsplit, scratch.data_ptr(), scratch.numel(), _sp()), "rows")
    # fused route
    k1, v1 = base_k.clone(), base_v.clone()
    o1 = torch.zeros(rows, NQ, dtype=torch.float16, device=dev)
    sq, sk, sv = _tile_partials(uq), _tile_partials(uk), _tile_partials(uv)
    L.check(lib.onebit_attention_decode_rows_fused(uq.data_ptr(), uk.data_ptr(), uv.data_ptr(), sq.data_ptr(), sk.data_ptr(), sv.data_ptr(), bp(bq), bp(bk),
                                                   bp(bv), cos.data_ptr(), sin.data_ptr(), k1.data_ptr(), v1.data_ptr(), o1.data_ptr(), hn.data_ptr(),
                                                   row_slot.data_ptr(), row_pos.data_ptr(), rows, H, Hkv, D, slots, max_len, max_len, chunk, nsplit, 1e-5,
                                                   scratch.data_ptr(), scratch.numel(), _sp()), "fused")
    torch.cuda.synchronize()
    live = slice(0, rows - 1)
    assert float(o1[-1].abs().max()) == 0.0                                        # idle row untouched
    scale = float(o0[live].float().abs().max())
    assert float((o0[live].float() - o1[live].float()).abs().max()) <= 3e-3 * scale
    assert float((k0.float() - k1.float()).abs().max()) <= 2.0 ** -9 * float(k0.abs().max())
    assert float((v0.float() - v1.float()).abs().max()) <= 2.0 ** -9 * float(v0.abs().max())
    assert float((k0 != k1).float().mean()) < 0.01
    changed = (k1 != base_k).any(-1).sum().item()                                   # exactly one row per (live request, kv head) was appended
    assert changed == (rows - 1) * Hkv


def test_mixed_step_full_size_properties():
    """BASELINE config 5's step at its real width and row count (LLaMA2-13B widths, 2 layers; 8 x 512 prompt tokens + 24 decode rows = 4120
    rows) through size-independent properties: (1) the step is deterministic and idempotent (the same items twice: bit-identical logits --
    fixed-order combines, tickets back at zero, the same cache rows rewritten); (2) the ORDER of the items does not matter: every item's
    logits are bit-identical under a permutation (a row's sums never mix with other rows': the GEMM's token columns, the row kernels, the
    ragged attention's segments are independent); (3) a prompt entering as one 512-token chunk or as 256 + 256 gives the same greedy token
    and logits within fp16 tolerance (the attention's key blocks split differently)."""
    from onebit_amd.engine import MixedStep
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device(DEV)
    cfg = OneBitLlamaConfig(vocab_size=1024, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2, num_attention_heads=40,
                            max_position_embeddings=1024)
    model = build_synthetic_model(cfg, seed=5, device=dev)
    slots, max_len = 32, 700
    g = torch.Generator().manual_seed(2)
    rnd = lambda n: torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist()

    def fresh():
        shape = (slots, cfg.num_key_value_heads, max_len, cfg.head_dim)
        caches = [(torch.zeros(shape, dtype=torch.float16, device=dev), torch.zeros(shape, dtype=torch.float16, device=dev)) for _ in range(cfg.num_hidden_layers)]
        return MixedStep(model, caches, slots, max_len, max_rows=4200, keep_logits=True)
    hist = [(s, 0, rnd(100 + 3 * s)) for s in range(24)]
    items = [(s, 100 + 3 * s, rnd(1)) for s in range(24)] + [(24 + i, 0, rnd(512)) for i in range(8)]
    ms = fresh()
    ms.launch(hist)
    n1 = ms.launch(items).clone()
    l1 = ms.logits[:len(items)].clone()
    n2 = ms.launch(items).clone()
    l2 = ms.logits[:len(items)].clone()
    torch.cuda.synchronize()
    assert torch.equal(l1, l2) and torch.equal(n1, n2)                                  # (1)
    assert torch.isfinite(l1.float()).all() and len(set(n1.tolist())) > 4
    perm = torch.randperm(len(items), generator=g).tolist()
    ms2 = fresh()
    ms2.launch(hist)
    n3 = ms2.launch([items[j] for j in perm]).clone()
    l3 = ms2.logits[:len(items)].clone()
    torch.cuda.synchronize()
    for pos_, j in enumerate(perm):                                                     # (2)
        assert torch.equal(l3[pos_], l1[j]) and int(n3[pos_]) == int(n1[j]), (pos_, j)
    ms3 = fresh()
    ms3.launch(hist)
    toks = items[24][2]
    ms3.launch([(24, 0, toks[:256])])
    n4 = ms3.launch([(24, 256, toks[256:])]).clone()
    l4 = ms3.logits[0].float().clone()
    torch.cuda.synchronize()
    ref = l1[24].float()
    assert float((l4 - ref).abs().max()) < 1e-2 * float(ref.abs().max())               # (3)
    srt = ref.sort().values
    if float(srt[-1] - srt[-2]) > 2e-2 * float(ref.abs().max()):
        assert int(n4[0]) == int(n1[24])


def test_mixed_step_randomised_mixes():
    """tools/mixed_fuzz.py, short form: random mixes of decoding slots and prompt chunks whose row counts cross every route threshold
    (one skinny launch / passes / grouped GEMM with or without a skinny tail / K-slices), five model shapes incl. GQA, attention_bias,
    7B and 13B widths: every item's logits and greedy token against the module path on the item's whole history."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mixed_fuzz.py"), "5", "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FUZZ ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
