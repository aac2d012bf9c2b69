"""GPU model-level parity: the host LLaMA around BitLinearInf against logits recorded from the
reference's BitLlamaForCausalLMInf on tiny configs (tests/golden/gen_goldens_model.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load_model(golden_dir, name, dev):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    cfg = OneBitLlamaConfig(**kw)
    model = OneBitLlamaForCausalLM(cfg, torch.float16)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    model.load_state_dict(sd)              # reference key layout loads unchanged (strict)
    return z, cfg, model.to(dev).eval()


@pytest.mark.parametrize("name", ["a", "b"])
def test_eager_model_matches_reference_logits(golden_dir, name):
    dev = torch.device("cuda:0")
    z, cfg, model = _load_model(golden_dir, name, dev)
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    cache = model.new_cache(1, 32)
    logits = model(ids, cache).cpu().numpy()
    ref16, ref32 = z["prefill_logits_f16"], z["prefill_logits_f32"]
    assert logits.shape == ref16.shape and logits.dtype == np.float32
    scale = np.abs(ref32).max()
    # the reference's own fp16-vs-fp32 gap sets the scale of what "parity" can mean here
    ref_gap = np.abs(ref16 - ref32).max()
    assert np.abs(logits - ref16).max() <= max(2.0 * ref_gap, 2e-3 * scale), (np.abs(logits - ref16).max(), ref_gap)
    assert np.abs(logits - ref32).max() <= max(2.0 * ref_gap, 2e-3 * scale)
    # incremental decode with the KV cache, feeding the reference's greedy tokens
    toks = torch.from_numpy(z["greedy_f16"]).to(dev)
    dec = []
    for i in range(4):
        dec.append(model(toks[:, i:i + 1], cache).cpu().numpy())
    dec = np.concatenate(dec, axis=1)
    assert np.abs(dec - z["decode_logits_f16"]).max() <= max(2.0 * ref_gap, 2e-3 * scale)
    # greedy tokens agree wherever the reference's own top-2 margin is not within noise
    out = model.generate(ids, max_new_tokens=5)[:, ids.shape[1]:].cpu().numpy()
    margin = z["margin_f16"]
    for i in range(5):
        if margin[:i + 1].min() > 4.0 * max(ref_gap, 1e-3):
            assert out[0, i] == z["greedy_f16"][0, i], (i, out, z["greedy_f16"])


def test_convert_train_checkpoint_and_serve(tmp_path):
    """Converter step + checkpoint round trip + fused engine: latent weights -> packed inference
    checkpoint on disk (reference layout) -> load -> logits identical to the in-memory model."""
    from onebit_amd import checkpoint as C
    from onebit_amd.engine import DecodeEngine
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM, synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=96, hidden_size=128, intermediate_size=384, num_hidden_layers=2,
                            num_attention_heads=2, max_position_embeddings=64)
    sd = synthetic_state_dict(cfg, seed=11)
    g = torch.Generator().manual_seed(3)
    train_sd = {}
    for k, v in sd.items():                      # fabricate latent weights whose signs are the packed bits
        if v.dtype == torch.int8:
            bits = (v.view(torch.uint8)[:, :, None] >> torch.arange(8, dtype=torch.uint8)) & 1
            signs = (1.0 - 2.0 * bits.reshape(v.shape[0], -1).float())
            train_sd[k] = signs * (0.01 + torch.rand(signs.shape, generator=g))
        else:
            train_sd[k] = v
    conv = C.convert_train_state_dict(train_sd, device=dev)
    for k, v in sd.items():
        assert torch.equal(conv[k].cpu(), v), k
    model = OneBitLlamaForCausalLM(cfg, torch.float16)
    model.load_state_dict(conv)
    C.save_inference_checkpoint(model, str(tmp_path))
    served = C.load_inference_checkpoint(str(tmp_path), device=dev)
    ids = torch.tensor([[5, 9, 2, 77]], device=dev)
    ref = model.to(dev).eval()(ids)
    assert torch.equal(served(ids), ref)
    out = DecodeEngine(served, max_len=32).generate(ids, max_new_tokens=4)
    assert out.shape == (1, 8)


def test_sdpa_prefill_attention_within_fp16_tolerance_of_eager(golden_dir):
    """The optional fused prefill attention (model.set_attention("sdpa")) against the eager op order
    and against the reference's recorded prefill logits, same bar as the eager test."""
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(golden_dir, "model_tiny_a.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    model = model.to(dev).eval()
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    eager = model(ids).cpu().numpy()
    fused = model.set_attention("sdpa")(ids).cpu().numpy()
    model.set_attention("eager")
    ref16, ref32 = z["prefill_logits_f16"], z["prefill_logits_f32"]
    tol = max(2.0 * np.abs(ref16 - ref32).max(), 2e-3 * np.abs(ref32).max())
    assert np.abs(fused - ref16).max() <= tol
    assert np.abs(fused - eager).max() <= tol


@pytest.mark.parametrize("name", ["a", "b"])
def test_fused_glue_forward_within_fp16_tolerance(golden_dir, name):
    """model.set_fused_glue(): pre-LayerNorm outputs + onebit_rows_res_ln_rms / onebit_rows_swiglu, prefill
    and incremental decode with a cache, against the reference's recorded logits (same bar as the default path)."""
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    model = model.to(dev).eval().set_fused_glue(True)
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    ref16, ref32 = z["prefill_logits_f16"], z["prefill_logits_f32"]
    tol = max(2.0 * np.abs(ref16 - ref32).max(), 2e-3 * np.abs(ref32).max())
    cache = model.new_cache(1, ids.shape[1] + 4)
    lg = model(ids, cache).cpu().numpy()
    assert np.abs(lg - ref16).max() <= tol
    toks = z["greedy_f16"][0]
    lg2 = model(torch.tensor([[int(toks[0]), int(toks[1])]], device=dev), cache).cpu().numpy()   # 2 tokens on top of the cache
    assert np.abs(lg2[0, 0] - z["decode_logits_f16"][0][0]).max() <= tol
    assert np.abs(lg2[0, 1] - z["decode_logits_f16"][0][1]).max() <= tol


def test_tensor_parallel_prefill_world1_matches_module_path():
    """onebit_amd/tp.py with the HIP callbacks at tensor-parallel degree 1 (the only degree a 1-GPU box
    offers; degree 2 runs on CPU over gloo in tests/test_tp_cpu.py): K-sharded o / down through
    onebit_matmul_partial + onebit_scale_layernorm, N-sharded q|k|v / gate|up through the row-statistics
    kernels, against the module path's logits."""
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    from onebit_amd.tp import TensorParallelPrefill
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=3,
                            num_attention_heads=8, num_key_value_heads=4, max_position_embeddings=128)
    model = build_synthetic_model(cfg, seed=6, device=dev)
    ids = torch.randint(0, 512, (3, 37), generator=torch.Generator().manual_seed(2)).to(dev)
    ref = model(ids)
    tp = TensorParallelPrefill(model, 0, 1)
    got = tp(ids)
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 6e-3 * scale
    assert tp.exchanges == 2 * cfg.num_hidden_layers
    assert tp.kv[0][0].shape == (3, 4, 37, cfg.head_dim)


def test_fused_tensor_parallel_prefill_world1_and_stats_kernels(golden_dir):
    """The fused tensor-parallel route (tp.py with HipGlue: onebit_rows_qkv_rope_stats / onebit_rows_swiglu_stats with the
    statistics GIVEN, onebit_rows_res_ln_rms on the reduce-scattered rows) at degree 1: against the module path on a
    GQA model, against the REFERENCE's recorded logits on the tiny golden model, and the two *_stats kernels against
    their self-computing forms fed the same statistics."""
    import ctypes
    from onebit_amd import _lib
    from onebit_amd.bitnet import _stream_ptr
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    from onebit_amd.tp import TensorParallelPrefill
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=3,
                            num_attention_heads=8, num_key_value_heads=4, max_position_embeddings=128)
    model = build_synthetic_model(cfg, seed=6, device=dev)
    ids = torch.randint(0, 512, (3, 37), generator=torch.Generator().manual_seed(2)).to(dev)
    ref = model(ids)
    scale = float(ref.abs().max())
    for impl in ("sdpa", "hip"):
        tp = TensorParallelPrefill(model, 0, 1, attention=impl)
        assert tp.fused
        got = tp(ids)
        assert float((got - ref).abs().max()) <= 6e-3 * scale, impl
    assert tp.exchanges == 2 * cfg.num_hidden_layers and tp.kv[0][0].shape == (3, 4, 37, cfg.head_dim)
    assert not TensorParallelPrefill(model, 0, 1).fused              # attention="eager": torch glue, reference op order
    # the reference's own logits (tiny golden model b: every projection on the MFMA path)
    z = np.load(os.path.join(golden_dir, "model_tiny_b.npz"))
    from onebit_amd.llama import OneBitLlamaForCausalLM
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    gm = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    gm.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    gm = gm.to(dev).eval()
    lg = TensorParallelPrefill(gm, 0, 1, attention="hip")(torch.from_numpy(z["input_ids"]).to(dev)).cpu().numpy()
    ref16, ref32 = z["prefill_logits_f16"], z["prefill_logits_f32"]
    assert np.abs(lg - ref16).max() <= max(2.0 * np.abs(ref16 - ref32).max(), 2e-3 * np.abs(ref32).max())
    # *_stats kernels == the self-computing kernels when given the statistics those compute (fp32 two-pass here: the
    # outputs may differ by an fp16 ulp where mean / rstd differ in the last fp32 bits)
    lib, sp = _lib.load(), _stream_ptr(dev)
    T, I = 19, 1408
    g = torch.Generator().manual_seed(4)
    ug = (0.4 * torch.randn(T, I, generator=g) + 0.1).half().to(dev)
    uu = (0.6 * torch.randn(T, I, generator=g)).half().to(dev)
    def mr(u):
        f = u.float()
        m = f.mean(-1)
        return m, torch.rsqrt(((f - m[:, None]) ** 2).mean(-1) + 1e-5)
    st4 = torch.stack([*mr(ug), *mr(uu)], dim=-1).contiguous()
    a0, a1 = torch.empty_like(ug), torch.empty_like(ug)
    _lib.check(lib.onebit_rows_swiglu(ug.data_ptr(), uu.data_ptr(), None, a0.data_ptr(), T, I, 1e-5, sp), "swiglu")
    _lib.check(lib.onebit_rows_swiglu_stats(ug.data_ptr(), uu.data_ptr(), None, st4.data_ptr(), a1.data_ptr(), T, I, 1e-5, sp), "swiglu_stats")
    d = (a0.float() - a1.float()).abs()
    assert float(d.max()) <= 2.0 ** -9 * max(1.0, float(a0.float().abs().max())) and float((d > 0).float().mean()) <= 0.02


def test_rows_qkv_rope_matches_module_ops():
    """onebit_rows_qkv_rope (LayerNorm of the q|k|v rows + RoPE + head transpose, k / v into cache rows at
    past_len) against the module path's torch ops (F.layer_norm, the rotate_half formula of
    modeling_bitllama.py:175-181 with every op rounded to fp16), grouped-query shape, batch 3, past 5."""
    from onebit_amd import _lib
    from onebit_amd.bitnet import _stream_ptr
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, S, Hh, Hkv, D, past, max_len, max_pos = 3, 21, 8, 2, 64, 5, 40, 64
    g = torch.Generator().manual_seed(11)
    u_q = (0.3 * torch.randn(B * S, Hh * D, generator=g) + 0.05).half().to(dev)
    u_k = (0.2 * torch.randn(B * S, Hkv * D, generator=g) - 0.1).half().to(dev)
    u_v = (0.5 * torch.randn(B * S, Hkv * D, generator=g)).half().to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(max_pos).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    cos, sin = emb.cos().half().to(dev), emb.sin().half().to(dev)
    q = torch.zeros(B, Hh, S, D, dtype=torch.float16, device=dev)
    kc = torch.zeros(B + 1, Hkv, max_len, D, dtype=torch.float16, device=dev)
    vc = torch.zeros_like(kc)
    rc = lib.onebit_rows_qkv_rope(u_q.data_ptr(), u_k.data_ptr(), u_v.data_ptr(), cos.data_ptr(), sin.data_ptr(), q.data_ptr(),
                                  kc.data_ptr(), vc.data_ptr(), B, S, Hh, Hkv, D, past, max_len, max_pos, 1e-5, 0, _stream_ptr(dev))
    _lib.check(rc, "rows_qkv_rope")
    q_tm = torch.zeros(B, S, Hh, D, dtype=torch.float16, device=dev)        # ONEBIT_FLAG_Q_TOKEN_MAJOR: same values, [B, S, H, D]
    rc = lib.onebit_rows_qkv_rope(u_q.data_ptr(), u_k.data_ptr(), u_v.data_ptr(), cos.data_ptr(), sin.data_ptr(), q_tm.data_ptr(),
                                  kc.data_ptr(), vc.data_ptr(), B, S, Hh, Hkv, D, past, max_len, max_pos, 1e-5, 0x2, _stream_ptr(dev))
    _lib.check(rc, "rows_qkv_rope")
    assert torch.equal(q_tm.transpose(1, 2), q)

    def rot(x):
        return torch.cat((-x[..., D // 2:], x[..., :D // 2]), dim=-1)
    c, s = cos[past:past + S][None, None], sin[past:past + S][None, None]
    lq = torch.nn.functional.layer_norm(u_q, (Hh * D,)).view(B, S, Hh, D).transpose(1, 2)
    lk = torch.nn.functional.layer_norm(u_k, (Hkv * D,)).view(B, S, Hkv, D).transpose(1, 2)
    lv = torch.nn.functional.layer_norm(u_v, (Hkv * D,)).view(B, S, Hkv, D).transpose(1, 2)
    q_ref = (lq * c) + (rot(lq) * s)
    k_ref = (lk * c) + (rot(lk) * s)
    # the LayerNorm statistics are fp32 sums in a different order: a handful of elements move by one fp16 ulp
    for got, ref in ((q, q_ref), (kc[:B, :, past:past + S], k_ref), (vc[:B, :, past:past + S], lv)):
        d = (got.float() - ref.float()).abs()
        ulp = torch.clamp(ref.float().abs(), min=2.0 ** -14) * 2.0 ** -10
        assert float((d / ulp).max()) <= 2.001
        assert float((d > 0).float().mean()) <= 0.02
    # nothing outside rows [past, past + S) of slots [0, B) was touched
    assert float(kc[B].abs().max()) == 0 and float(kc[:, :, :past].abs().max()) == 0 and float(kc[:, :, past + S:].abs().max()) == 0
    # error behaviour: tokens beyond the cache
    assert lib.onebit_rows_qkv_rope(u_q.data_ptr(), u_k.data_ptr(), u_v.data_ptr(), cos.data_ptr(), sin.data_ptr(), q.data_ptr(),
                                    kc.data_ptr(), vc.data_ptr(), B, S, Hh, Hkv, D, 30, max_len, max_pos, 1e-5, 0, _stream_ptr(dev)) != 0


@pytest.mark.parametrize("name", ["a", "b"])
def test_fused_glue_with_sdpa_uses_rows_qkv_rope_within_tolerance(golden_dir, name):
    """set_fused_glue + set_attention("sdpa"): the prefill route bench.py's prefill_model field times
    (q|k|v through onebit_rows_qkv_rope into the cache, fused causal attention) against the reference's
    recorded prefill logits, then two decode tokens on top of the cache that route filled."""
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    model = model.to(dev).eval().set_fused_glue(True).set_attention("sdpa")
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    ref16, ref32 = z["prefill_logits_f16"], z["prefill_logits_f32"]
    tol = max(2.0 * np.abs(ref16 - ref32).max(), 2e-3 * np.abs(ref32).max())
    cache = model.new_cache(1, ids.shape[1] + 4)
    lg = model(ids, cache).cpu().numpy()
    assert np.abs(lg - ref16).max() <= tol
    toks = z["greedy_f16"][0]
    lg2 = model(torch.tensor([[int(toks[0]), int(toks[1])]], device=dev), cache).cpu().numpy()
    assert np.abs(lg2[0, 0] - z["decode_logits_f16"][0][0]).max() <= tol
    assert np.abs(lg2[0, 1] - z["decode_logits_f16"][0][1]).max() <= tol


def test_prescaled_prefill_route_forced_on_a_small_model():
    """The fused forward's pre-scaled route (producers write fp16(x * h), projections run with
    ONEBIT_FLAG_PRESCALED) only opens where the LDS-DMA GEMM is eligible; OB_GEMM3=2 forces that on a small
    model in a child interpreter, which compares the route with the module path's logits."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "forced_prescaled_child.py")
    r = subprocess.run([sys.executable, child], env=dict(os.environ, OB_GEMM3="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forced-prescaled ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_rows_glue_scaled_outputs_small_ragged_shapes():
    """onebit_rows_res_ln_rms with 0..3 scaled outputs and onebit_rows_swiglu with h_next on widths that are not
    multiples of the 4096-element pass (H = 40, 1000, 5120): x and the residual are unchanged by the extra
    outputs, every a_i equals fp16(x * h_i) exactly; argument errors are reported."""
    import ctypes
    from onebit_amd import _lib
    from onebit_amd.bitnet import _stream_ptr
    lib = _lib.load()
    dev = torch.device("cuda:0")
    sp = _stream_ptr(dev)
    g = torch.Generator().manual_seed(17)
    nul = (ctypes.c_void_p * 3)()
    for T, H in ((3, 40), (7, 1000), (2, 5120)):
        hres = torch.randn(T, H, generator=g).half().to(dev)
        u = (0.4 * torch.randn(T, H, generator=g) + 0.1).half().to(dev)
        w = (1.0 + 0.2 * torch.randn(H, generator=g)).half().to(dev)
        hs = [(0.1 * (0.5 + torch.rand(H, generator=g))).half().to(dev) for _ in range(3)]
        hout0, x0 = torch.empty_like(hres), torch.empty_like(hres)
        _lib.check(lib.onebit_rows_res_ln_rms(hres.data_ptr(), u.data_ptr(), w.data_ptr(), hout0.data_ptr(), x0.data_ptr(),
                                              nul, nul, 0, T, H, 1e-6, 1e-5, sp), "rows_res_ln_rms")
        for n in (1, 2, 3):
            hout, xs = torch.empty_like(hres), [torch.empty_like(hres) for _ in range(n)]
            hp = (ctypes.c_void_p * 3)(*[h.data_ptr() for h in hs[:n]])
            xp = (ctypes.c_void_p * 3)(*[a.data_ptr() for a in xs])
            _lib.check(lib.onebit_rows_res_ln_rms(hres.data_ptr(), u.data_ptr(), w.data_ptr(), hout.data_ptr(), None,
                                                  hp, xp, n, T, H, 1e-6, 1e-5, sp), "rows_res_ln_rms")
            assert torch.equal(hout, hout0)
            for a, h in zip(xs, hs):
                assert torch.equal(a, x0 * h)
        ug, uu = (0.5 * torch.randn(T, H, generator=g)).half().to(dev), (0.5 * torch.randn(T, H, generator=g)).half().to(dev)
        act, act_s = torch.empty_like(ug), torch.empty_like(ug)
        _lib.check(lib.onebit_rows_swiglu(ug.data_ptr(), uu.data_ptr(), None, act.data_ptr(), T, H, 1e-5, sp), "rows_swiglu")
        _lib.check(lib.onebit_rows_swiglu(ug.data_ptr(), uu.data_ptr(), hs[0].data_ptr(), act_s.data_ptr(), T, H, 1e-5, sp), "rows_swiglu")
        assert torch.equal(act_s, act * hs[0])
    # errors: neither x nor a scaled output; too many scaled outputs; a NULL entry in the list
    assert lib.onebit_rows_res_ln_rms(hres.data_ptr(), u.data_ptr(), w.data_ptr(), hout0.data_ptr(), None, nul, nul, 0,
                                      T, H, 1e-6, 1e-5, sp) != 0
    assert lib.onebit_rows_res_ln_rms(hres.data_ptr(), u.data_ptr(), w.data_ptr(), hout0.data_ptr(), x0.data_ptr(), nul, nul, 4,
                                      T, H, 1e-6, 1e-5, sp) != 0
    assert lib.onebit_rows_res_ln_rms(hres.data_ptr(), u.data_ptr(), w.data_ptr(), hout0.data_ptr(), x0.data_ptr(), nul, nul, 1,
                                      T, H, 1e-6, 1e-5, sp) != 0


@pytest.mark.parametrize("kind", ["linear", "dynamic", "none"])
def test_rope_scaling_variants_match_reference(golden_dir, kind):
    """The reference's RoPE scaling variants (modeling_bitllama.py:123-165: linear = positions / factor, dynamic NTK = base
    rescaled once the sequence exceeds max_position_embeddings) and its on-demand growth of the rotary cache: a
    24-token prompt on a model with max_position_embeddings = 16 (so "dynamic" rescales, twice more during the two
    cached decode steps, exactly as the reference's stateful cache does), fp32 parameters, against logits recorded
    from the reference model (tests/golden/gen_goldens_rope.py)."""
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM, synthetic_state_dict
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(golden_dir, "model_rope.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    rs = None if kind == "none" else {"type": kind, "factor": 2.0}
    cfg = OneBitLlamaConfig(rope_scaling=rs, **kw)
    model = OneBitLlamaForCausalLM(cfg, torch.float32)
    sd = {k: (v if v.dtype == torch.int8 else v.float()) for k, v in synthetic_state_dict(OneBitLlamaConfig(**kw), seed=7, dtype=torch.float16).items()}
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    ref = z["logits_" + kind]
    cache = model.new_cache(1, 32)
    got = [model(ids, cache).cpu().numpy()]
    toks = z["greedy_" + kind]
    for i in range(2):
        got.append(model(torch.from_numpy(toks[:, i:i + 1]).to(dev), cache).cpu().numpy())
    got = np.concatenate(got, axis=1)
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), np.abs(got - ref).max()
    if kind != "none":                                       # and the variants really differ from the unscaled model
        assert np.abs(z["logits_" + kind] - z["logits_none"]).max() > 50 * np.abs(got - ref).max()
    with pytest.raises(ValueError):
        OneBitLlamaConfig(rope_scaling={"type": "yarn", "factor": 2.0}, **kw)
    with pytest.raises(ValueError):
        OneBitLlamaConfig(rope_scaling={"type": "linear", "factor": 1}, **kw)
