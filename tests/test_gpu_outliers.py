"""Adversarial dynamic range on the fused decode GEMV launches (round-4 review, weak #2).

The integer-digit path (ob_decode.h, MATH == 1) quantises a = fp16(x * h) to 23-bit fixed point PER WAVE CHUNK (512
elements) relative to that chunk's largest element: exact for every element within 12 binades of the chunk maximum, the
rest rounded at 2^-23 of it.  Every other GPU test feeds N(0, 1) activations, where that never matters.  Trained
LLaMA-family residual streams carry outlier channels 10^3 .. 10^4 x the median, so here the input of each of the four
prologues (PLAIN, EMBED_RMS, RES_LN_RMS, SWIGLU; 7B and 13B shapes) carries 8 channels at 2^10 .. 2^15 x the median:

  * +A / -A pairs inside ONE 512-element chunk with equal input_factor (rows whose two sign bits agree cancel the pair
    exactly),
  * +B / -B pairs in two chunks owned by different waves (the cancellation happens in the cross-wave fp32 sum);
  * all eight channels are paired, so 1/16 of the rows cancel EVERY outlier and are left with the sum of the
    ~1e-4-times-smaller rest, quantised at the outliers' scale -- the worst case of a per-chunk exponent.

Bar (reference semantics: bitnet.py:113-116 = an fp16 GEMM with fp32 accumulation, the order unspecified): pre-LayerNorm u
within 2 fp16 ulps of the oracle's exactly-summed value PLUS the noise any fp32 accumulation of K such addends has,
|g_n| * 1.0 U with U = max|a| * 2^-23 * sqrt(K) -- sequential fp32 summation of K addends has sigma = 0.29 U, i.e. 4 sigma =
1.2 U; a priori the digit path (quantisation step <= 2 max|a| 2^-23 on at most K contaminated elements) is bounded by
4 sigma <= 2.3 U, measured (profiles/r05_outlier_parity.txt) its worst excess over the 2 rounding ulps is 0.25 U, so the bar
is set BELOW what a sequential fp32 accumulation is entitled to.  Rows where the largest pair does not cancel are dominated
by the outliers and must meet the plain 2-ulp bar.  The measured errors go to gpurun_out/ when OB_WRITE_PROFILES=1.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FP16_ULP = 2.0 ** -10
BAR_U = 1.0          # accumulation allowance in units of U = max|a| 2^-23 sqrt(K); measured worst: 0.25 (profiles/r05_outlier_parity.txt)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REPORT = []


def _mk(K, N, seed, dev, tie=()):
    from onebit_amd import BitLinearInf
    rng = np.random.default_rng(seed)
    packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
    flip = lambda n: np.where(rng.random(n) < 0.1, -1.0, 1.0)
    h = (0.1 * (0.5 + rng.random(K)) * flip(K)).astype(np.float16)
    for a, b in tie:                       # equal input_factor on a pair: +A and -A cancel exactly after the scaling
        h[b] = h[a]
    g = (0.1 * (0.5 + rng.random(N)) * flip(N)).astype(np.float16)
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.from_numpy(packed).to(dev)
    m.input_factor.data = torch.from_numpy(h).to(dev)
    m.weight_scale.data = torch.from_numpy(g).to(dev)
    return m, packed, h, g


# (index, multiple of the median).  FOUR cancelling pairs -- 1/16 of the rows agree in sign on all of them and are left with
# the sum of the N(0, 1) rest, quantised at the outliers' scale: pair inside chunk 0 (wave 0), pair across chunks 1 and 2
# (waves 1 and 2: cancels in the cross-wave sum), pair inside chunk 3, pair across chunks 4 and 5
_PAIRS = ((5, 300), (512 + 17, 1024 + 400), (1536 + 164, 1536 + 364), (2048 + 252, 2560 + 340))
_OUTLIERS = ((5, 2.0 ** 14), (300, -2.0 ** 14), (512 + 17, 2.0 ** 15), (1024 + 400, -2.0 ** 15),
             (1536 + 164, -2.0 ** 10), (1536 + 364, 2.0 ** 10), (2048 + 252, 2.0 ** 12), (2560 + 340, -2.0 ** 12))


def _outlier_vec(n, rng, top=None):
    """N(0, 1) with the outlier channels planted; `top` caps the largest magnitude (fp16 range of later products)."""
    x = rng.standard_normal(n).astype(np.float32)
    med = float(np.median(np.abs(x)))
    scale = 1.0 if top is None else min(1.0, top / (2.0 ** 15 * med))
    x *= scale
    for i, mult in _OUTLIERS:
        x[i] = mult * med * scale
    return x.astype(np.float16)


def _check_u_outliers(got, ref, a, g, K, tag):
    """2 fp16 ulps + |g| * BAR_U * U (module docstring); returns the measured figures for the profile."""
    got, ref = got.astype(np.float64), ref.astype(np.float64)
    amax = float(np.abs(a.astype(np.float64)).max())
    U = amax * 2.0 ** -23 * np.sqrt(K)
    ulp = np.maximum(np.abs(ref), 2.0 ** -12) * FP16_ULP
    err = np.abs(got - ref)
    allow = 2.001 * ulp + np.abs(g.astype(np.float64)) * BAR_U * U
    worst_ulp = float((err / ulp).max())
    # error left after the two rounding ulps, in units of |g| U (what the accumulation itself contributes)
    worst_U = float((np.maximum(err - 2.001 * ulp, 0.0) / (np.abs(g.astype(np.float64)) * U + 1e-30)).max())
    frac2 = float((err > 2.001 * ulp).mean())
    _REPORT.append(f"{tag:<44s} max|a| {amax:9.2f}  U {U:.3e}  worst {worst_ulp:6.2f} ulp  beyond 2 ulp: {100 * frac2:5.2f} % of rows, "
                   f"worst excess {worst_U:5.2f} |g|U  (bar {BAR_U})")
    assert (err <= allow).all(), (tag, worst_ulp, worst_U)
    return worst_ulp, frac2, worst_U


def _cancel_rows(packed, pairs):
    """rows whose sign bits agree on every planted pair (the pairs cancel there)"""
    bits = np.unpackbits(packed.view(np.uint8), axis=1, bitorder="little")
    ok = np.ones(packed.shape[0], dtype=bool)
    for a, b in pairs:
        ok &= bits[:, a] == bits[:, b]
    return ok


@pytest.mark.parametrize("H,I", [(4096, 11008), (5120, 13824)])
def test_fused_gemv_outlier_activations_vs_oracle(coracle, H, I):
    from onebit_amd.engine import PRO_EMBED_RMS, PRO_PLAIN, PRO_RES_LN_RMS, PRO_SWIGLU, fused_gemv, tile_stats_floats
    dev = torch.device("cuda:0")
    f16 = torch.float16
    rng = np.random.default_rng(77)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = lambda n: torch.full((tile_stats_floats(n),), float("nan"), dtype=torch.float32, device=dev)
    tag = lambda s: f"{'7B' if H == 4096 else '13B'} {s}"

    # ---- 1. PLAIN (o_proj): the outlier vector is the input itself -------------------------------------------------
    o_m, o_w, o_h, o_g = _mk(H, H, 1, dev, tie=_PAIRS)
    x = _outlier_vec(H, rng)
    u_o = torch.empty(H, dtype=f16, device=dev)
    st_o = st(H)
    fused_gemv([o_m], [u_o], PRO_PLAIN, xin=t(x), stats_out=[st_o])
    _, ref = coracle.forward_f16(o_w, x[None], o_h, o_g, None, return_pre_ln=True)
    a = (x * o_h).astype(np.float16)
    assert a[5] == -a[300] and a[512 + 17] == -a[1024 + 400]          # the planted pairs do cancel after the scaling
    got = u_o.cpu().numpy()
    _check_u_outliers(got, ref[0], a, o_g, H, tag("PLAIN o_proj"))
    # rows where no pair cancels are dominated by the outliers: plain 2-ulp bar, no allowance
    nc = ~_cancel_rows(o_w, _PAIRS[1:2])                                # the 2^15 pair does not cancel there
    ulp = np.maximum(np.abs(ref[0].astype(np.float32)), 2.0 ** -12) * FP16_ULP
    assert (np.abs(got.astype(np.float32) - ref[0].astype(np.float32))[nc] <= 2.001 * ulp[nc]).all()

    # ---- 2. EMBED_RMS (first q|k|v): outliers in the embedding row ------------------------------------------------------
    q_m, q_w, q_h, q_g = _mk(H, H, 2, dev, tie=_PAIRS)
    k_m, k_w, k_h, k_g = _mk(H, H, 3, dev, tie=_PAIRS)
    v_m, v_w, v_h, v_g = _mk(H, H, 4, dev, tie=_PAIRS)
    emb = np.stack([rng.standard_normal(H).astype(np.float16), _outlier_vec(H, rng), rng.standard_normal(H).astype(np.float16)])
    rms_w = np.ones(H, dtype=np.float16)        # (a learned RMSNorm weight would un-tie the pairs; the dynamic range is the point)
    token = torch.tensor([1], dtype=torch.int32, device=dev)
    u_q, u_k, u_v = (torch.empty(H, dtype=f16, device=dev) for _ in range(3))
    hout = torch.empty(H, dtype=f16, device=dev)
    fused_gemv([q_m, k_m, v_m], [u_q, u_k, u_v], PRO_EMBED_RMS, embed=t(emb), token=token, hres_out=hout, rms_w=t(rms_w),
               stats_out=[st(H), st(H), st(H)])
    assert np.array_equal(hout.cpu().numpy(), emb[1])
    def rmsnorm(hk):
        hk = hk.astype(np.float32)
        rs = 1.0 / np.sqrt((hk.astype(np.float64) ** 2).mean() + 1e-6)
        return (rms_w * (hk * rs).astype(np.float16)).astype(np.float16)
    xn = rmsnorm(emb[1])
    for m_, w_, h_, g_, u_, nm in ((q_m, q_w, q_h, q_g, u_q, "q"), (k_m, k_w, k_h, k_g, u_k, "k"), (v_m, v_w, v_h, v_g, u_v, "v")):
        _, ref = coracle.forward_f16(w_, xn[None], h_, g_, None, return_pre_ln=True)
        _check_u_outliers(u_.cpu().numpy(), ref[0], (xn * h_).astype(np.float16), g_, H, tag("EMBED_RMS " + nm))

    # ---- 3. RES_LN_RMS (gate | up): outliers in the residual stream, N(0, 1) in the projection output it adds ----------
    g_m, g_w, g_h, g_g = _mk(H, I, 5, dev, tie=_PAIRS)
    p_m, p_w, p_h, p_g = _mk(H, I, 6, dev, tie=_PAIRS)
    hres = _outlier_vec(H, rng)
    u_prev = rng.standard_normal(H).astype(np.float16)
    st_prev = st(H)
    # tile partials of u_prev as a producer would have published them
    up64 = u_prev.astype(np.float64).reshape(-1, 16)
    sp = np.zeros(tile_stats_floats(H), dtype=np.float32)
    sp[: 2 * (H // 16)].reshape(-1, 2)[:, 0] = up64.sum(1)
    sp[: 2 * (H // 16)].reshape(-1, 2)[:, 1] = ((up64 - up64.mean(1, keepdims=True)) ** 2).sum(1)
    st_prev.copy_(torch.from_numpy(sp))
    u_gate, u_up = torch.empty(I, dtype=f16, device=dev), torch.empty(I, dtype=f16, device=dev)
    st_g, st_u = st(I), st(I)
    for use_stats in (True, False):
        kw = dict(st_prev=st_prev) if use_stats else {}
        fused_gemv([g_m, p_m], [u_gate, u_up], PRO_RES_LN_RMS, hres_in=t(hres), u_prev=t(u_prev), hres_out=hout, rms_w=t(rms_w),
                   stats_out=[st_g, st_u], **kw)
        xn_k = rmsnorm(hout.cpu().numpy())                     # the residual stream the kernel formed (isolates the GEMV)
        for w_, h_, g_, u_, nm in ((g_w, g_h, g_g, u_gate, "gate"), (p_w, p_h, p_g, u_up, "up")):
            _, ref = coracle.forward_f16(w_, xn_k[None], h_, g_, None, return_pre_ln=True)
            _check_u_outliers(u_.cpu().numpy(), ref[0], (xn_k * h_).astype(np.float16), g_, H, tag(f"RES_LN_RMS {nm} stats={int(use_stats)}"))

    # ---- 4. SWIGLU (down): outliers in the pre-LayerNorm gate and up rows --------------------------------------------------
    d_m, d_w, d_h, d_g = _mk(I, H, 7, dev, tie=_PAIRS)
    # SiLU removes negative gate values, so the planted SIGNS ride on the up row and the gate carries the same channels
    # positive: act = silu(LN(gate)) * LN(up) then has +-pairs that cancel up to the fp16 rounding of the two LayerNorms
    uu = _outlier_vec(I, rng)
    ug = rng.standard_normal(I).astype(np.float16)
    ug[list(i for i, _ in _OUTLIERS)] = np.abs(uu[list(i for i, _ in _OUTLIERS)])
    for v_, s_ in ((ug, st_g), (uu, st_u)):
        v64 = v_.astype(np.float64).reshape(-1, 16)
        sp = np.zeros(tile_stats_floats(I), dtype=np.float32)
        sp[: 2 * (I // 16)].reshape(-1, 2)[:, 0] = v64.sum(1)
        sp[: 2 * (I // 16)].reshape(-1, 2)[:, 1] = ((v64 - v64.mean(1, keepdims=True)) ** 2).sum(1)
        s_.copy_(torch.from_numpy(sp))
    # the activation from the device's own row kernel (onebit_rows_swiglu: the arithmetic of this prologue -- hardware exp2 / rcp,
    # fma_mix LayerNorm -- once per row), so that the oracle is fed what the GEMV multiplies: a host-side restatement lands one
    # fp16 ulp away on ~1e-3 of the elements, and one ulp of an outlier channel is hundreds of ulps of a row that cancels
    from onebit_amd import _lib
    from onebit_amd.bitnet import _stream_ptr
    act_d = torch.empty(1, I, dtype=f16, device=dev)
    ugd, uud = t(ug[None]), t(uu[None])
    _lib.check(_lib.load().onebit_rows_swiglu(ugd.data_ptr(), uud.data_ptr(), None, act_d.data_ptr(), 1, I, 1e-5, _stream_ptr(dev)), "rows_swiglu")
    act = act_d.cpu().numpy()[0]
    def lnv(v):
        v = v.astype(np.float32)
        return ((v - v.mean(dtype=np.float64)) / np.sqrt(v.var(dtype=np.float64) + 1e-5)).astype(np.float16)
    gl32 = lnv(ug).astype(np.float32)
    act_host = ((gl32 / (1.0 + np.exp(-gl32.astype(np.float64)))).astype(np.float16) * lnv(uu)).astype(np.float16)
    assert np.isfinite(act.astype(np.float32)).all()
    assert (act != act_host).mean() <= 0.01                      # the row kernel against the plain statement: rounding ties only
    assert np.abs(act.astype(np.float32) - act_host.astype(np.float32)).max() <= 2 * FP16_ULP * np.abs(act_host.astype(np.float32)).max()
    _, ref = coracle.forward_f16(d_w, act[None], d_h, d_g, None, return_pre_ln=True)
    a = (act * d_h).astype(np.float16)
    u_down = torch.empty(H, dtype=f16, device=dev)
    for use_stats in (True, False):
        kw = dict(st_gate=st_g, st_up=st_u) if use_stats else {}
        fused_gemv([d_m], [u_down], PRO_SWIGLU, u_gate=t(ug), u_up=t(uu), **kw)
        got = u_down.cpu().numpy()
        # the two statistics forms (tile partials / recomputed) round mean and rstd differently in the last fp32 bit: an element of
        # the prologue's activation may sit one fp16 ulp from the row kernel's; when that element is an outlier channel every row
        # moves by |g| * 2^-10 |a|_max.  Allowed once (one element), reported
        amax = float(np.abs(a.astype(np.float64)).max())
        err = np.abs(got.astype(np.float64) - ref[0].astype(np.float64))
        ulp = np.maximum(np.abs(ref[0].astype(np.float64)), 2.0 ** -12) * FP16_ULP
        U = amax * 2.0 ** -23 * np.sqrt(I)
        flip = np.abs(d_g.astype(np.float64)) * amax * FP16_ULP
        strict = (err <= 2.001 * ulp + np.abs(d_g.astype(np.float64)) * BAR_U * U).all()
        _REPORT.append(f"{tag(f'SWIGLU down stats={int(use_stats)}'):<44s} max|a| {amax:9.2f}  U {U:.3e}  worst {float((err / ulp).max()):6.2f} ulp  "
                       f"beyond 2 ulp: {100 * float((err > 2.001 * ulp).mean()):5.2f} % of rows, worst excess "
                       f"{float((np.maximum(err - 2.001 * ulp, 0.0) / (np.abs(d_g.astype(np.float64)) * U + 1e-30)).max()):5.2f} |g|U  (bar {BAR_U})"
                       f"{'' if strict else '  [one activation element differs from the row kernel by an fp16 ulp: allowed]'}")
        assert (err <= 2.001 * ulp + np.abs(d_g.astype(np.float64)) * BAR_U * U + flip).all(), (use_stats, float((err / ulp).max()))
        rel = np.linalg.norm(got.astype(np.float32) - ref[0].astype(np.float32)) / np.linalg.norm(ref[0].astype(np.float32))
        assert rel <= 1e-3, rel

    if os.environ.get("OB_WRITE_PROFILES") == "1":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r05_outlier_parity.txt"), "a") as f:
            f.write("\n".join(_REPORT) + "\n")
        _REPORT.clear()
