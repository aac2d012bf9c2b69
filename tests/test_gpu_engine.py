"""GPU parity of the fused decode engine (onebit_decode_step): against the reference's recorded
decode logits (tiny config), and against the module path on larger shapes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _golden_model(golden_dir, name, dev):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    return z, model.to(dev).eval()


@pytest.mark.parametrize("use_graph", [False, True])
def test_engine_matches_reference_decode_logits(golden_dir, use_graph):
    from onebit_amd.engine import DecodeEngine
    dev = torch.device("cuda:0")
    z, model = _golden_model(golden_dir, "b", dev)
    eng = DecodeEngine(model, max_len=32, use_graph=use_graph)
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    eng.prefill(ids)
    ref16, ref32 = z["decode_logits_f16"][0], z["decode_logits_f32"][0]
    gap = np.abs(ref16 - ref32).max()
    tol = max(2.0 * gap, 2e-3 * np.abs(ref32).max())
    toks = z["greedy_f16"][0]
    for i in range(4):
        eng.set_state(int(toks[i]), ids.shape[1] + i)      # feed the reference's tokens
        eng.step()
        lg = eng.logits().cpu().numpy()
        assert np.abs(lg - ref16[i]).max() <= tol, (i, np.abs(lg - ref16[i]).max(), tol)
        assert int(eng.pos.item()) == ids.shape[1] + i + 1
        assert int(eng.token.item()) == int(lg.argmax())


def test_engine_rejects_unsupported_shapes(golden_dir):
    from onebit_amd.engine import DecodeEngine
    dev = torch.device("cuda:0")
    _, model = _golden_model(golden_dir, "a", dev)          # intermediate 688: K % 32 != 0
    with pytest.raises(ValueError):
        DecodeEngine(model, max_len=32)


@pytest.mark.parametrize("cfgkw,steps", [
    (dict(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=3,
          num_attention_heads=8, max_position_embeddings=128), 24),
    # 7B-shaped layers (hidden 4096 / inter 11008 / head_dim 128), 2 layers
    (dict(vocab_size=4000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=2,
          num_attention_heads=32, max_position_embeddings=96), 12),
    # 13B-shaped layer: K = 5120 (10 chunks) and 13824 (27 chunks)
    (dict(vocab_size=1000, hidden_size=5120, intermediate_size=13824, num_hidden_layers=1,
          num_attention_heads=40, max_position_embeddings=64), 6),
])
def test_engine_matches_module_path(cfgkw, steps):
    from onebit_amd.engine import DecodeEngine
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(**cfgkw)
    model = build_synthetic_model(cfg, seed=3, device=dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    ids = torch.randint(0, cfg.vocab_size, (1, 9), generator=g).to(dev)
    # module path, token by token
    cache = model.new_cache(1, cfg.max_position_embeddings)
    lg = model(ids, cache)
    tok = lg[:, -1].argmax(-1, keepdim=True)
    ref_logits, ref_toks = [], [int(tok)]
    for _ in range(steps):
        lg = model(tok, cache)
        ref_logits.append(lg[0, -1].cpu().numpy())
        tok = lg[:, -1].argmax(-1, keepdim=True)
        ref_toks.append(int(tok))
    outs = {}
    for use_graph in (False, True):
        eng = DecodeEngine(model, max_len=cfg.max_position_embeddings, use_graph=use_graph)
        eng.prefill(ids)
        assert eng.first_token == ref_toks[0]
        got = []
        for i in range(steps):
            eng.set_state(ref_toks[i], ids.shape[1] + i)    # teacher-forced: compare logits step by step
            eng.step()
            got.append(eng.logits().cpu().numpy())
        outs[use_graph] = np.stack(got)
        ref = np.stack(ref_logits)
        scale = np.abs(ref).max()
        err = np.abs(outs[use_graph] - ref).max()
        # same rounding points as the module path; attention/softmax use fp32 inside -> small fp16-level noise
        assert err <= 6e-3 * scale, (use_graph, err, scale)
    np.testing.assert_array_equal(outs[False], outs[True])   # graph replay is bitwise the direct launch
    # free-running greedy generation agrees wherever the module path's top-2 margin is clear
    eng = DecodeEngine(model, max_len=cfg.max_position_embeddings, use_graph=True)
    out = eng.generate(ids, max_new_tokens=steps)[0, ids.shape[1]:].tolist()
    ref = np.stack(ref_logits)
    for i in range(steps):
        assert out[i] == ref_toks[i], (i, out, ref_toks)
        srt = np.sort(ref[i])
        if i < steps - 1 and srt[-1] - srt[-2] < 8e-3 * np.abs(ref).max():
            break                                            # margin within noise: later tokens may diverge


def _ref_u(mod, x):
    """pre-LayerNorm output of a BitLinearInf through the module path (its own HIP kernel, already
    pinned against the oracle): u = fp16(fp16(W.(h*x)) * g)."""
    ln, mod.layernorm = mod.layernorm, torch.nn.Identity()
    try:
        return mod(x[None])[0]
    finally:
        mod.layernorm = ln


@pytest.mark.parametrize("H,I", [(4096, 11008), (5120, 13824), (512, 1408), (128, 352)])
def test_fused_gemv_each_prologue(H, I):
    """onebit_fused_gemv launch by launch (the building blocks of onebit_decode_step) against the same
    math assembled from torch fp16 ops in the reference's order + the module-path BitLinearInf."""
    from onebit_amd.engine import PRO_EMBED_RMS, PRO_PLAIN, PRO_RES_LN_RMS, PRO_SWIGLU, fused_gemv
    from onebit_amd.llama import LlamaRMSNorm, OneBitLlamaConfig, build_synthetic_model
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=64, hidden_size=H, intermediate_size=I, num_hidden_layers=1,
                            num_attention_heads=H // 64, max_position_embeddings=16)
    model = build_synthetic_model(cfg, seed=5, device=dev)
    layer = model.model.layers[0]
    g = torch.Generator(device="cpu").manual_seed(0)
    f16 = torch.float16
    rnd = lambda n, s=1.0: (s * torch.randn(n, generator=g)).to(f16).to(dev)
    rms_w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(f16).to(dev)
    norm = LlamaRMSNorm(H, eps=1e-6, dtype=f16).to(dev)
    norm.weight.data = rms_w
    ln = lambda v: F.layer_norm(v, (v.numel(),), eps=1e-5)

    def check(got, ref, tag):
        got, ref = got.float(), ref.float()
        ulp = ref.abs().clamp_min(2.0 ** -14) * 2.0 ** -10
        bad = ((got - ref).abs() > 2.001 * ulp)
        # inputs of the GEMV may differ by an fp16 ulp in a few elements (rsqrt / exp approximations),
        # which moves every output slightly: bound the aggregate instead of each element
        rel = (got - ref).norm() / ref.norm()
        gross = ((got - ref).abs() > 0.02 * ref.abs().clamp_min(0.1)).nonzero().flatten()
        assert rel <= 1e-3, (tag, float(rel), int(gross.numel()), gross[:32].tolist())
        assert bad.float().mean() <= 0.05, (tag, float(bad.float().mean()))

    a, mlp = layer.self_attn, layer.mlp
    # PLAIN: o_proj
    x = rnd(H)
    out = torch.empty(H, device=dev, dtype=f16)
    fused_gemv([a.o_proj], [out], PRO_PLAIN, xin=x)
    check(out, _ref_u(a.o_proj, x), "plain")
    # RES_LN_RMS: q, k, v (3 projections) and gate, up (2)
    hres, u_prev = rnd(H), rnd(H, 3.0) + 0.7
    r = hres + ln(u_prev)
    xn = norm(r)
    outs = [torch.empty(H, device=dev, dtype=f16) for _ in range(3)]
    hout = torch.empty(H, device=dev, dtype=f16)
    fused_gemv([a.q_proj, a.k_proj, a.v_proj], outs, PRO_RES_LN_RMS, hres_in=hres, u_prev=u_prev, hres_out=hout, rms_w=rms_w)
    # the residual stream written by workgroup 0: torch's LayerNorm statistics may differ in the last fp32 bit
    assert (hout.float() - r.float()).abs().max() <= 2.0 ** -9 * max(1.0, float(r.abs().max()))
    assert (hout != r).float().mean() <= 0.01
    for o, m, n in zip(outs, (a.q_proj, a.k_proj, a.v_proj), "qkv"):
        check(o, _ref_u(m, xn), "res_ln_rms " + n)
    og, ou = torch.empty(I, device=dev, dtype=f16), torch.empty(I, device=dev, dtype=f16)
    fused_gemv([mlp.gate_proj, mlp.up_proj], [og, ou], PRO_RES_LN_RMS, hres_in=hres, u_prev=u_prev, hres_out=hout, rms_w=rms_w)
    check(og, _ref_u(mlp.gate_proj, xn), "gate")
    check(ou, _ref_u(mlp.up_proj, xn), "up")
    # EMBED_RMS
    tok = torch.tensor([7], device=dev, dtype=torch.int32)
    emb = model.model.embed_tokens.weight
    fused_gemv([a.q_proj, a.k_proj, a.v_proj], outs, PRO_EMBED_RMS, embed=emb, token=tok, hres_out=hout, rms_w=rms_w)
    assert torch.equal(hout, emb[7])
    check(outs[0], _ref_u(a.q_proj, norm(emb[7])), "embed q")
    check(outs[2], _ref_u(a.v_proj, norm(emb[7])), "embed v")
    # SWIGLU: down
    ug, uu = rnd(I, 2.0) - 0.3, rnd(I, 0.5) + 0.2
    act = F.silu(ln(ug)) * ln(uu)
    od = torch.empty(H, device=dev, dtype=f16)
    fused_gemv([mlp.down_proj], [od], PRO_SWIGLU, u_gate=ug, u_up=uu)
    check(od, _ref_u(mlp.down_proj, act), "swiglu")


@pytest.mark.parametrize("cfgkw,prompt_len,steps,engine_kw", [
    # grouped-query attention: 8 query heads share 2 kv heads
    (dict(vocab_size=256, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8,
          num_key_value_heads=2, max_position_embeddings=64), 7, 10, dict(long_context_from=0)),
    # long context: > 256 cached positions (keys beyond the register-preloaded window, values beyond 128)
    (dict(vocab_size=256, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
          max_position_embeddings=512), 300, 6, dict(long_context_from=0)),
    # the same through the split-KV attention (4 splits of 128 positions: 3 hold data at 300 tokens)
    (dict(vocab_size=256, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
          max_position_embeddings=512), 300, 6, dict(long_context_from=64, attn_splits=4, long_attention="pair")),
    # ... and through the key-block attention (round 6: rope / append launch + onebit_attention_decode_rows, 64 positions per split)
    (dict(vocab_size=256, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
          max_position_embeddings=512), 300, 6, dict(long_context_from=64, long_attention="keyblock", attn_chunk=64)),
    # positions 55 .. 74: the first nine steps replay the graph whose attention requests 64 positions before it knows the
    # position (onebit_decode_state_t.attn_blind), the rest the 128-position graph -- the same logits on either side of 64
    (dict(vocab_size=256, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
          max_position_embeddings=128), 55, 20, dict(long_context_from=0)),
    # split-KV with grouped-query attention and a position crossing a split boundary (chunk = 32)
    (dict(vocab_size=256, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8,
          num_key_value_heads=2, max_position_embeddings=256), 60, 10, dict(long_context_from=16, attn_splits=8, long_attention="pair")),
    # key-block form with grouped-query attention, positions 60 .. 69 crossing the 64-position split boundary
    (dict(vocab_size=256, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8,
          num_key_value_heads=2, max_position_embeddings=256), 60, 10, dict(long_context_from=16, long_attention="keyblock", attn_chunk=64)),
])
def test_engine_attention_variants(cfgkw, prompt_len, steps, engine_kw):
    from onebit_amd.engine import DecodeEngine
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(**cfgkw)
    model = build_synthetic_model(cfg, seed=9, device=dev)
    g = torch.Generator(device="cpu").manual_seed(2)
    ids = torch.randint(0, cfg.vocab_size, (1, prompt_len), generator=g).to(dev)
    cache = model.new_cache(1, cfg.max_position_embeddings)
    lg = model(ids, cache)
    tok = lg[:, -1].argmax(-1, keepdim=True)
    ref_logits, ref_toks = [], [int(tok)]
    for _ in range(steps):
        lg = model(tok, cache)
        ref_logits.append(lg[0, -1].cpu().numpy())
        tok = lg[:, -1].argmax(-1, keepdim=True)
        ref_toks.append(int(tok))
    eng = DecodeEngine(model, max_len=cfg.max_position_embeddings, **engine_kw)
    assert (eng.graph_long is not None or bool(eng._kb_graphs)) == bool(engine_kw.get("long_context_from"))
    assert bool(eng._kb_graphs) == (bool(engine_kw.get("long_context_from")) and engine_kw.get("long_attention") == "keyblock")
    assert eng.graph64 is not None and eng._state64.attn_blind == 64 and eng._state.attn_blind == 0
    eng.prefill(ids)
    assert eng.first_token == ref_toks[0]
    ref = np.stack(ref_logits)
    for i in range(steps):
        eng.set_state(ref_toks[i], prompt_len + i)
        eng.step()
        err = np.abs(eng.logits().cpu().numpy() - ref[i]).max()
        assert err <= 6e-3 * np.abs(ref).max(), (i, err)


def test_engine_is_deterministic_across_graphs_and_runs():
    """Two engines, 200 greedy tokens each, crossing from the one-workgroup-per-head attention graph
    into the split-KV graph (last-arriver combine in fixed split order): identical token streams."""
    from onebit_amd.engine import DecodeEngine
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=3,
                            num_attention_heads=8, max_position_embeddings=320)
    model = build_synthetic_model(cfg, seed=4, device=dev)
    ids = torch.randint(0, 512, (1, 40), generator=torch.Generator().manual_seed(1)).to(dev)
    runs = []
    for _ in range(2):
        eng = DecodeEngine(model, max_len=320, long_context_from=96, attn_splits=4, long_attention="pair")
        runs.append(eng.generate(ids, 200)[0].tolist())
    assert runs[0] == runs[1]
    kb = []
    for _ in range(2):                           # the key-block form: fixed-order combine by the last arriver, tickets back at zero
        eng = DecodeEngine(model, max_len=320, long_context_from=96, long_attention="keyblock", attn_chunk=64)
        kb.append(eng.generate(ids, 200)[0].tolist())
    assert kb[0] == kb[1]
    assert len(set(runs[0][40:])) > 8            # not a degenerate constant stream


def test_fused_gemv_propagates_non_finite_activations():
    """The integer sign path quantises the activations; an Inf / NaN among them must still poison the
    outputs the way it does in the reference's fp16 GEMM (every output row sums over all of K)."""
    from onebit_amd import BitLinearInf
    from onebit_amd.engine import PRO_PLAIN, fused_gemv
    dev = torch.device("cuda:0")
    K, N = 8192, 256                                   # two 4096-chunks: the integer path is taken
    g = torch.Generator().manual_seed(0)
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(K, generator=g).half().to(dev)
    out = torch.empty(N, dtype=torch.float16, device=dev)
    fused_gemv([m], [out], PRO_PLAIN, xin=x)
    assert torch.isfinite(out).all()
    for bad in (float("nan"), float("inf")):
        xb = x.clone()
        xb[5000] = bad
        fused_gemv([m], [out], PRO_PLAIN, xin=xb)
        assert torch.isnan(out).all()


def test_engine_accepts_fp32_checkpoint(golden_dir):
    """The released OneBit checkpoints are FP32 (/root/reference/checkpoints/README.md:10).  The fused engines cast the
    floating parameters once at build -- what from_pretrained(torch_dtype=float16) does (modeling_utils.py:696) --
    sharing the packed int8 weights, and leave the caller's fp32 model untouched: an engine built from the fp32 model
    gives the logits of one built from the fp16 model, bit for bit, and they meet the reference's fp16 decode logits."""
    from onebit_amd.engine import BatchedDecodeStep, DecodeEngine, fp16_view
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(golden_dir, "model_tiny_b.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    m32 = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float32)
    m32.load_state_dict({k: (v if v.dtype == torch.int8 else v.float()) for k, v in sd.items()})
    m32 = m32.to(dev).eval()
    m16 = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    m16.load_state_dict(sd)
    m16 = m16.to(dev).eval()
    view = fp16_view(m32)
    assert view is not m32 and fp16_view(m16) is m16
    q32, qv = m32.model.layers[0].self_attn.q_proj, view.model.layers[0].self_attn.q_proj
    assert qv.weight.data_ptr() == q32.weight.data_ptr()                   # packed weights shared, not copied
    assert q32.weight_scale.dtype == torch.float32 and qv.weight_scale.dtype == torch.float16
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    toks = z["greedy_f16"][0]
    ref16, ref32 = z["decode_logits_f16"][0], z["decode_logits_f32"][0]
    tol = max(2.0 * np.abs(ref16 - ref32).max(), 2e-3 * np.abs(ref32).max())
    got = {}
    for name, model in (("fp32", m32), ("fp16", m16)):
        eng = DecodeEngine(model, max_len=32)
        eng.prefill(ids)
        out = []
        for i in range(3):
            eng.set_state(int(toks[i]), ids.shape[1] + i)
            eng.step()
            out.append(eng.logits().cpu().numpy())
        got[name] = np.stack(out)
        assert np.abs(got[name] - ref16[:3]).max() <= tol
    assert np.array_equal(got["fp32"], got["fp16"])
    assert m32.lm_head.weight.dtype == torch.float32                       # the caller's model was not modified
    # the batched step takes the fp32 model the same way
    cache = view.new_cache(2, 16)
    BatchedDecodeStep(m32, cache.layers, 2, 16)
    # ... and refuses the fp32 model's OWN caches (fp32 storage would be read and written as fp16 by the attention kernels)
    with pytest.raises(ValueError, match="float16"):
        BatchedDecodeStep(m32, m32.new_cache(2, 16).layers, 2, 16)


@pytest.mark.parametrize("route", ["module", "engine", "engine graph", "engine split-kv", "engine key-block", "batched", "batched no-stats",
                                   "batched unscaled", "batched key-block", "fused prefill", "mixed step"])
def test_attention_bias_checkpoint_vs_reference(golden_dir, route):
    """config.attention_bias = True (modeling_bitllama.py:451-454: q / k / v / o_proj are built with a bias, added after the
    projection's LayerNorm, bitnet.py:119-120).  Round 4 refused such checkpoints in the fused engines; now the attention
    kernels add b_q / b_k / b_v before the rotary embedding and the gate|up prologue (batched: the post-attention row kernel)
    adds b_o to LayerNorm(u_o).  Logits of every route against tests/golden/model_tiny_bias.npz, recorded from the
    reference's BitLlamaForCausalLMInf with attention_bias=True (gen_goldens_model.py bias; 3 layers)."""
    from onebit_amd.engine import BatchedDecodeStep, DecodeEngine
    dev = torch.device("cuda:0")
    z, model = _golden_model(golden_dir, "bias", dev)
    assert model.model.layers[0].self_attn.q_proj.bias is not None and model.model.layers[0].self_attn.o_proj.bias is not None
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    S = ids.shape[1]
    ref16, ref32 = z["decode_logits_f16"][0], z["decode_logits_f32"][0]
    tol = max(2.0 * float(np.abs(ref16 - ref32).max()), 2e-3 * float(np.abs(ref32).max()))
    toks = z["greedy_f16"][0]
    # the bias is not a no-op in this fixture: the same weights without it miss the reference by far more than the bar
    if route == "module":
        cache = model.new_cache(1, 32)
        lg = model(ids, cache).cpu().numpy()
        assert np.abs(lg - z["prefill_logits_f16"]).max() <= tol
        tk = torch.from_numpy(z["greedy_f16"]).to(dev)
        dec = np.concatenate([model(tk[:, i:i + 1], cache).cpu().numpy() for i in range(4)], axis=1)
        assert np.abs(dec[0] - ref16).max() <= tol
        for layer in model.model.layers:
            for p_ in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj):
                p_.bias = None
        cache0 = model.new_cache(1, 32)
        model(ids, cache0)
        dec0 = model(tk[:, 0:1], cache0).cpu().numpy()
        assert np.abs(dec0[0, 0] - ref16[0]).max() > 4 * tol
        return
    if route == "fused prefill":
        # round 6: the fused prefill glue takes the biases too (q / k / v in the ragged rope kernel, o in onebit_rows_res_ln_rms_bias)
        model.set_attention("hip").set_fused_glue(True)
        cache = model.new_cache(1, 32)
        lg = model(ids, cache).cpu().numpy()
        assert np.abs(lg - z["prefill_logits_f16"]).max() <= tol
        tk = torch.from_numpy(z["greedy_f16"]).to(dev)
        dec = np.concatenate([model(tk[:, i:i + 1], cache).cpu().numpy() for i in range(4)], axis=1)
        assert np.abs(dec[0] - ref16).max() <= tol
        return
    if route == "mixed step":
        # round 6: onebit_mixed_step -- the prompt as one chunk in slot 2, then teacher-forced single-token steps next to a second request
        from onebit_amd.engine import MixedStep
        cfg = model.config
        shape = (3, cfg.num_key_value_heads, 32, cfg.head_dim)
        caches = [(torch.zeros(shape, dtype=torch.float16, device=dev), torch.zeros(shape, dtype=torch.float16, device=dev)) for _ in range(cfg.num_hidden_layers)]
        ms = MixedStep(model, caches, 3, 32, keep_logits=True)
        nxt = ms.launch([(2, 0, ids[0].tolist())])
        torch.cuda.synchronize()
        assert np.abs(ms.logits[0].float().cpu().numpy() - z["prefill_logits_f16"][0, -1]).max() <= tol and int(nxt[0]) == int(toks[0])
        for i in range(4):
            ms.launch([(0, 0, ids[0, :5].tolist()), (2, S + i, [int(toks[i])])] if i == 1 else [(2, S + i, [int(toks[i])])])
            torch.cuda.synchronize()
            lg = ms.logits[1 if i == 1 else 0].float().cpu().numpy()
            assert np.abs(lg - ref16[i]).max() <= tol, (i, float(np.abs(lg - ref16[i]).max()), tol)
        return
    if route.startswith("engine"):
        kw = dict(use_graph=route == "engine graph")
        if route == "engine split-kv":
            kw.update(long_context_from=4, attn_splits=2, long_attention="pair")        # every step on the two split-KV launches
        if route == "engine key-block":
            kw.update(long_context_from=4, long_attention="keyblock", attn_chunk=64)    # rope / append launch + key-block attention
        eng = DecodeEngine(model, max_len=64, **kw)
        eng.prefill(ids)
        assert eng.first_token == int(toks[0])
        for i in range(4):
            eng.set_state(int(toks[i]), S + i)                   # teacher-forced with the reference's tokens
            eng.step()
            lg = eng.logits().cpu().numpy()
            assert np.abs(lg - ref16[i]).max() <= tol, (route, i, float(np.abs(lg - ref16[i]).max()), tol)
            assert np.abs(lg - ref32[i]).max() <= tol
        return
    # batched step: three slots decode the same sequence (every slot against the reference), one idle slot in between
    B, max_len = 4, 32
    cache = model.new_cache(B, max_len)
    model(ids.repeat(B, 1), cache)
    step = BatchedDecodeStep(model, cache.layers, B, max_len, sample=True, keep_logits=True,
                             producer_stats=route != "batched no-stats", prescaled_rows=route != "batched unscaled",
                             attn_splits=1 if route == "batched key-block" else 0, attn_chunk=64)
    for i in range(4):
        step.tokens.fill_(int(toks[i]))
        step.pos.fill_(S + i)
        step.pos[2] = -1                                          # idle slot: computed, never appended
        step.launch()
        torch.cuda.synchronize()
        got = step.logits.float().cpu().numpy()
        for b_ in (0, 1, 3):
            assert np.abs(got[b_] - ref16[i]).max() <= tol, (route, i, b_, float(np.abs(got[b_] - ref16[i]).max()), tol)
        assert (step.next_tokens.cpu().numpy()[[0, 1, 3]] == got[[0, 1, 3]].argmax(-1)).all()


@pytest.mark.parametrize("K,Ns", [
    (4096, (2048, 2048)),             # 2 x 128 tiles            -> 1 slot per workgroup
    (4096, (4096, 4096)),             # 2 x 256                  -> 2
    (4096, (6144, 6144)),             # 2 x 384                  -> 3
    (4096, (10240, 10240)),           # 2 x 640                  -> 5
    (4096, (16384, 16384)),           # 2 x 1024                 -> 8
    (4096, (4096, 1024, 1024)),       # grouped-query shapes: 256 + 64 + 64 tiles -> 2, unequal workgroup ranges
    (8192, (8192, 8192, 8192)),       # two 512-weight chunks per wave (KV = 2), 3 x 512 tiles -> 6
    (8192, (14336, 14336)),           # KV = 2, 2 x 896 -> 7
    (2048, (2072, 1048)),             # K below one chunk row (waves 4..7 own no chunk), ragged last tiles (N % 16 != 0: no partials)
])
@pytest.mark.parametrize("use_stats", [True, False])
def test_one_projection_per_workgroup_launches(coracle, K, Ns, use_stats):
    """The q|k|v / gate|up launches of the decode step deal their workgroups to the projections (ob_decode.h, WGP): every
    slot count the host can pick (1 .. 8), one and two chunks per wave, unequal projection heights, a chunk row that only
    half the waves populate and ragged last tiles -- each projection's pre-LayerNorm u against the oracle fed with the
    normalised vector the kernel itself formed (u within 2 fp16 ulps, different at all on < 3 % of the elements), and the
    per-tile LayerNorm partials it publishes against the stored u."""
    from onebit_amd import BitLinearInf
    from onebit_amd.engine import PRO_PLAIN, PRO_RES_LN_RMS, fused_gemv, tile_stats_floats
    dev = torch.device("cuda:0")
    f16 = torch.float16
    rng = np.random.default_rng(K + sum(Ns))
    t = lambda a: torch.from_numpy(a).to(dev)
    mods, raw = [], []
    for i, N in enumerate(Ns):
        packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
        flip = lambda n: np.where(rng.random(n) < 0.1, -1.0, 1.0)
        h = (0.1 * (0.5 + rng.random(K)) * flip(K)).astype(np.float16)
        g = (0.1 * (0.5 + rng.random(N)) * flip(N)).astype(np.float16)
        m = BitLinearInf(K, N, dtype=f16).to(dev)
        m.weight.data, m.input_factor.data, m.weight_scale.data = t(packed), t(h), t(g)
        mods.append(m); raw.append((packed, h, g))
    # a producer for u_prev and its partials: any PLAIN launch of width K
    prod = BitLinearInf(K, K, dtype=f16).to(dev)
    prod.weight.data = t(rng.integers(0, 256, (K, K // 8), dtype=np.uint8).view(np.int8))
    prod.input_factor.data = t((0.1 * (0.5 + rng.random(K))).astype(np.float16))
    prod.weight_scale.data = t((0.1 * (0.5 + rng.random(K))).astype(np.float16))
    u_prev = torch.empty(K, dtype=f16, device=dev)
    st_prev = torch.full((tile_stats_floats(K),), float("nan"), dtype=torch.float32, device=dev)
    fused_gemv([prod], [u_prev], PRO_PLAIN, xin=t(rng.standard_normal(K).astype(np.float16)), stats_out=[st_prev])
    hres = rng.standard_normal(K).astype(np.float16)
    rms_w = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float16)
    outs = [torch.empty(N, dtype=f16, device=dev) for N in Ns]
    sts = [torch.full((tile_stats_floats(N),), float("nan"), dtype=torch.float32, device=dev) for N in Ns]
    hout = torch.empty(K, dtype=f16, device=dev)
    kw = dict(st_prev=st_prev) if use_stats else {}
    fused_gemv(mods, outs, PRO_RES_LN_RMS, hres_in=t(hres), u_prev=u_prev, hres_out=hout, rms_w=t(rms_w), stats_out=sts, **kw)
    hk = hout.cpu().numpy().astype(np.float32)
    rs = 1.0 / np.sqrt((hk.astype(np.float64) ** 2).mean() + 1e-6)
    xn = (rms_w * (hk * rs).astype(np.float16)).astype(np.float16)
    for (packed, h, g), o, st, N in zip(raw, outs, sts, Ns):
        _, u_ref = coracle.forward_f16(packed, xn[None], h, g, None, return_pre_ln=True)
        got, ref = o.cpu().numpy().astype(np.float32), u_ref[0].astype(np.float32)
        ulp = np.maximum(np.abs(ref), 2.0 ** -12) * 2.0 ** -10
        # The oracle's input vector is rebuilt on the host from the residual stream the kernel wrote (exact rsqrt); the
        # kernel's hardware rsq may round ONE or two of the K normalised inputs to the neighbouring fp16 value.  Such a flip
        # moves EVERY z by up to one ulp of that input (2^-13 for |a| < 0.25), i.e. every u by up to 2^-13 * |g| -- invisible
        # against 2 ulps of a typical output, several ulps of an output that cancelled to ~1e-3.  Hence the absolute term
        # (two flips' worth) and the wider "different at all" fraction; a wrong tile, slot or projection is off by O(1).
        atol = 2.0 * 2.0 ** -13 * float(np.abs(g.astype(np.float32)).max())
        assert (np.abs(got - ref) <= 2.001 * ulp + atol).all(), (N, float((np.abs(got - ref) / ulp).max()))
        assert (got != ref).mean() <= 0.08, (N, float((got != ref).mean()))
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel <= 2e-4, (N, rel)
        if N % 16 == 0:
            uu = got.astype(np.float64).reshape(-1, 16)
            s = uu.sum(1)
            m2 = ((uu - s[:, None] / 16.0) ** 2).sum(1)
            pub = st.cpu().numpy()[: 2 * (N // 16)].reshape(-1, 2)
            np.testing.assert_allclose(pub[:, 0], s, rtol=1e-5, atol=1e-4)
            np.testing.assert_allclose(pub[:, 1], m2, rtol=1e-4, atol=1e-4)


def test_attn_blind_hint_does_not_change_results():
    """onebit_decode_state_t.attn_blind = 64 is a performance hint: the attention launch requests 64 instead of 128 cached
    positions before it knows the position and streams the rest -- logits bit-identical to the 128-position window at
    positions below, at and far above 64 (direct launches, no graph), and an unknown value is refused."""
    import ctypes
    from onebit_amd import _lib
    from onebit_amd.engine import DecodeEngine
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=256, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                            max_position_embeddings=256)
    model = build_synthetic_model(cfg, seed=12, device=dev)
    ids = torch.randint(0, cfg.vocab_size, (1, 200), generator=torch.Generator().manual_seed(5)).to(dev)
    eng = DecodeEngine(model, max_len=256, use_graph=False, long_context_from=0)
    eng.prefill(ids)                                           # the cache holds 200 positions
    for pos in (5, 63, 64, 65, 130, 199):
        outs = []
        for b64 in (False, True):
            eng.set_state(7, pos)
            eng._launch(blind64=b64)
            torch.cuda.synchronize()
            outs.append(eng.buf["logits"].clone())
        assert torch.equal(outs[0], outs[1]), pos
    eng._state64.attn_blind = 32
    eng.set_state(7, 5)
    with pytest.raises(Exception, match="attn_blind"):
        eng._launch(blind64=True)
    eng._state64.attn_blind = 64


@pytest.mark.parametrize("cfgkw,S,rows", [
    (dict(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=512), 150, 2048),
    (dict(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=512), 150, 64),
    (dict(vocab_size=640, hidden_size=1024, intermediate_size=2816, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
          max_position_embeddings=512), 333, 128),
    (dict(vocab_size=512, hidden_size=512, intermediate_size=1536, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=256,
          attention_bias=True), 1, 2048),
    (dict(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=512), 129, 64),
])
def test_engine_prime_is_the_native_prefill(golden_dir, cfgkw, S, rows):
    """DecodeEngine.prime (round 6): the prompt through onebit_mixed_step -- whole, in chunks of `rows` tokens (64 / 128: chunks with past),
    a one-token prompt -- against DecodeEngine.prefill (module path): K / V cache rows equal to 2^-9 relative, the first token and the
    following greedy tokens equal up to a near-tie of the module path's logits; generate() takes the native route (asserted)."""
    from onebit_amd.engine import DecodeEngine
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(**cfgkw)
    model = build_synthetic_model(cfg, seed=41, device=dev)
    ids = torch.randint(0, cfg.vocab_size, (1, S), generator=torch.Generator().manual_seed(S)).to(dev)
    max_len = S + 12
    ref_eng = DecodeEngine(model, max_len=max_len, native_prefill=False)
    eng = DecodeEngine(model, max_len=max_len, prefill_rows=rows)
    lg = ref_eng.prefill(ids)[0, -1].float()
    first = eng.prime(ids)
    torch.cuda.synchronize()
    assert eng._mixed is not None and eng._mixed.launches == -(-S // max(rows, 64))
    scale = float(lg.abs().max())
    assert first == ref_eng.first_token or abs(float(lg[first] - lg[ref_eng.first_token])) < 2e-2 * scale
    for (ka, va), (kb, vb) in zip(eng.cache.layers, ref_eng.cache.layers):
        for a, b in ((ka, kb), (va, vb)):
            a, b = a[0, :, :S].float(), b[0, :, :S].float()
            assert float((a - b).abs().max()) <= 2.0 ** -8 * max(float(b.abs().max()), 1.0)
    got = eng.generate(ids, 10)[0, S:].tolist()
    ref = ref_eng.generate(ids, 10)[0, S:].tolist()
    assert ref_eng._mixed is None and len(got) == 10
    if got != ref:
        j = next(i for i in range(10) if got[i] != ref[i])
        l2 = model(torch.tensor([ids[0].tolist() + ref[:j]], device=dev))[0, -1].float()
        assert abs(float(l2[got[j]] - l2[ref[j]])) < 2e-2 * float(l2.abs().max()), (j, got, ref)


@pytest.mark.parametrize("B", [1, 3])
def test_generate_native_keeps_generates_contract(B):
    """OneBitLlamaForCausalLM.generate_native (DecodeEngine for B = 1, ContinuousBatcher for B > 1) against generate (module path):
    the same [B, S + n] tensor -- tokens equal up to a near-tie of the module path's logits --, with an EOS token that really occurs:
    rows end at their first EOS, continue with pad_token_id, the output ends where the last row ends."""
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=64, hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4,
                            max_position_embeddings=256)
    model = build_synthetic_model(cfg, seed=23, device=dev)
    ids = torch.randint(0, cfg.vocab_size, (B, 9), generator=torch.Generator().manual_seed(B)).to(dev)
    ref = model.generate(ids, 24)
    got = model.generate_native(ids, 24)
    assert got.shape == ref.shape and torch.equal(got[:, :9], ids)

    def same_or_near_tie(a, b):
        for r in range(B):
            ra, rb = a[r].tolist(), b[r].tolist()
            if ra != rb:
                j = next(i for i in range(len(ra)) if ra[i] != rb[i])
                lg = model(torch.tensor([rb[:j]], device=dev))[0, -1].float()
                assert abs(float(lg[ra[j]] - lg[rb[j]])) < 2e-2 * float(lg.abs().max()), (r, j)
                return False
        return True
    if same_or_near_tie(got, ref):
        eos = int(ref[0, 9 + 5])                                   # a token the first row really produces
        ref_e = model.generate(ids, 24, eos_token_id=eos, pad_token_id=0)
        got_e = model.generate_native(ids, 24, eos_token_id=eos, pad_token_id=0)
        assert torch.equal(got_e, ref_e)
        assert ref_e.shape[1] <= ref.shape[1] and (B > 1 or int(ref_e[0, -1]) == eos)
    assert len(model._native_engines) == 1                        # the engine is kept for the next call
