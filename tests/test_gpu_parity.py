"""GPU parity: the HIP path (through the C ABI) against the oracle and the reference's goldens.

Tolerance (fp16, BASELINE.json north_star: <= 1e-3 relative fp16 error): the kernels reproduce the
reference's fp16 rounding points, so against the oracle the pre-LayerNorm values agree to <= 2 fp16
ulps (accumulation-order flips of z, at most doubled by the *g rounding; < 2% of elements) and the outputs to rel-L2 <= 1e-3 / max-abs <= 2 fp16 ulps at
the output scale.  Integer work (pack / unpack) is bit-exact.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FP16_ULP = 2.0 ** -10


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from onebit_amd import _lib
    _lib.load()                      # native library must be present: no fallback
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _make_layer(K, N, dt, dev, packed, h, g, bias=None):
    from onebit_amd import BitLinearInf
    m = BitLinearInf(K, N, bias=bias is not None, dtype=dt).to(dev)
    m.weight.data = _t(packed, dev)
    m.input_factor.data = _t(h, dev).to(dt)
    m.weight_scale.data = _t(g, dev).to(dt)
    if bias is not None:
        m.bias.data = _t(bias, dev).to(dt)
    return m


def _check_f16(y, u, y_ref, u_ref, tag):
    y, u, y_ref, u_ref = (np.asarray(a, dtype=np.float32) for a in (y, u, y_ref, u_ref))
    # z flips by one fp16 ulp when the fp32 accumulation order straddles a rounding boundary; the
    # following fp16(z*g) can turn that into 2 ulps of u.  Anything beyond is a bug.
    ulp_u = np.maximum(np.abs(u_ref), 2.0 ** -14) * FP16_ULP
    bad_u = np.abs(u - u_ref) > 2.001 * ulp_u
    assert not bad_u.any(), (tag, "pre-LN differs by more than 2 fp16 ulps", int(bad_u.sum()))
    assert (u != u_ref).mean() <= 0.02, (tag, "too many ulp flips", float((u != u_ref).mean()))
    rel = np.linalg.norm(y - y_ref) / (np.linalg.norm(y_ref) + 1e-30)
    assert rel <= 1e-3, (tag, "rel-L2", rel)
    scale = max(1.0, float(np.abs(y_ref).max()))
    assert np.abs(y - y_ref).max() <= 2.5 * FP16_ULP * scale, (tag, float(np.abs(y - y_ref).max()))


def _golden_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "forward.npz"))
    meta = z["meta_idx_K_N_bias_nlead"]
    for row, dn, sp in zip(meta, z["meta_dtype"], z["meta_special"]):
        idx, K, N, bias, _ = (int(v) for v in row)
        p = f"c{idx}_"
        yield dict(idx=idx, K=K, N=N, dtype=str(dn), special=str(sp), packed=z[p + "packed"],
                   x=z[p + "x"], h=z[p + "h"], g=z[p + "g"],
                   bias=z[p + "bias"] if bias else None, y=z[p + "y"], u=z[p + "u"])


def test_goldens_forward(dev, golden_dir):
    """Every reference-generated forward fixture, fp32 and fp16, y and pre-LN u."""
    n = 0
    for c in _golden_cases(golden_dir):
        dt = torch.float16 if c["dtype"] == "f16" else torch.float32
        m = _make_layer(c["K"], c["N"], dt, dev, c["packed"], c["h"], c["g"], c["bias"])
        x = _t(c["x"], dev)
        y = m(x)
        assert y.shape == c["y"].shape and y.dtype == dt
        m.layernorm = torch.nn.Identity()
        b, m.bias = m.bias, None
        u = m(x)
        y, u = y.float().cpu().numpy(), u.float().cpu().numpy()
        if dt == torch.float32:
            su = np.abs(c["u"]).max() + 1e-30
            assert np.abs(u - c["u"]).max() <= 2e-5 * su, c["idx"]
            if c["special"] != "reset":
                assert np.abs(y - c["y"]).max() <= 2e-4, c["idx"]
        else:
            yr, ur = c["y"].astype(np.float32), c["u"].astype(np.float32)
            ulp_u = np.maximum(np.abs(ur), 2.0 ** -14) * FP16_ULP
            assert (np.abs(u - ur) <= 2.0 * ulp_u).all(), c["idx"]
            if c["special"] != "reset":
                rel = np.linalg.norm(y - yr) / np.linalg.norm(yr)
                assert rel <= 1e-3, (c["idx"], rel)
                assert np.abs(y - yr).max() <= 4e-3 * max(1.0, np.abs(yr).max()), c["idx"]
            else:
                assert np.abs(y - yr).max() <= 1e-3
        n += 1
    assert n == 32


SHAPES = [
    # (lead, K, N, bias)
    ((1,), 4096, 11008, False),       # BASELINE config 1 / 7B gate, up
    ((1,), 11008, 4096, False),       # 7B down (K = 21.5 x 512: tail path)
    ((1, 1), 4096, 4096, True),       # 7B q/k/v/o
    ((3,), 5120, 1024, False),        # 13B hidden
    ((2, 5), 256, 80, True),
    ((17,), 1376, 100, False),        # two token tiles, N not a multiple of 16
    ((33,), 512, 17, False),
    ((4,), 32, 1, False),             # single row, single word
    ((2, 100), 4096, 384, False),     # tiled MFMA GEMM (T > 16): 2 token tiles, 3 row tiles
    ((129,), 256, 130, True),         # GEMM with ragged token and row tiles
    ((40,), 352, 48, False),          # GEMM, K % 128 != 0 (partial last k-step)
    ((2,), 688, 48, False),           # K % 32 != 0 -> generic kernel
    ((5,), 40, 9, True),
]

# round 6: a few hundred rows of a projection whose tiles alone leave the chip idle -> K-sliced LDS-DMA GEMM (ob_gemm3_ksplit: fp32 sums per
# slice) + ONE row pass that adds the slices, applies fp16(fp16(.) * g) and the LayerNorm
KSLICED = [
    ((300,), 4096, 4096, True),       # 4 slices of 1024 columns, bias
    ((2, 257), 2816, 1000, False),    # 11 quads -> 2 slices (6 + 5), ragged token tile, N % 256 != 0
    ((200,), 11008, 4096, False),     # 43 quads -> 4 slices (11 + 11 + 11 + 10)
    ((70,), 4096, 4096, False),       # one token tile, 55 % full
]


@pytest.mark.parametrize("lead,K,N,bias", SHAPES)
def test_forward_fp16_vs_oracle(dev, coracle, lead, K, N, bias):
    rng = np.random.default_rng(K * 31 + N)
    packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
    x = rng.standard_normal((*lead, K)).astype(np.float16)
    flip = lambda n: np.where(rng.random(n) < 0.1, -1.0, 1.0)
    h = (0.1 * (0.5 + rng.random(K)) * flip(K)).astype(np.float16)
    g = (0.1 * (0.5 + rng.random(N)) * flip(N)).astype(np.float16)
    b = (0.1 * rng.standard_normal(N)).astype(np.float16) if bias else None
    y_ref, u_ref = coracle.forward_f16(packed, x, h, g, b, return_pre_ln=True)
    m = _make_layer(K, N, torch.float16, dev, packed, h, g, b)
    xt = _t(x, dev)
    y = m(xt)
    assert y.shape == (*lead, N) and y.dtype == torch.float16
    m.layernorm = torch.nn.Identity()
    m.bias = None
    u = m(xt)
    _check_f16(y.cpu().numpy(), u.cpu().numpy(), y_ref, u_ref, (lead, K, N))


@pytest.mark.parametrize("lead,K,N,bias", KSLICED)
def test_forward_fp16_k_sliced_route_vs_oracle(dev, coracle, lead, K, N, bias):
    """The module path at a few hundred rows (the K-sliced route) against the oracle.  Element-wise bar as in test_forward_fp16_vs_oracle with
    one difference: these cases hold 0.5-1.2 M outputs, some of which are sums that cancel to |z| << sigma_z = sqrt(K) * |a| -- there ANY two
    fp32 summation orders (the oracle's sequential one, four MFMA chains added at the end) differ by many ulps OF THE RESULT while agreeing to
    2^-20 of sigma_z.  So the 2-ulp bar holds for |u_ref| >= 2^-9 of the row's largest |u| and an absolute bar of 2 ulps of that fraction
    below it; the count of 1-ulp flips, the LayerNorm output and its rel-L2 are held as everywhere else."""
    rng = np.random.default_rng(K * 31 + N)
    packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
    x = rng.standard_normal((*lead, K)).astype(np.float16)
    flip = lambda n: np.where(rng.random(n) < 0.1, -1.0, 1.0)
    h = (0.1 * (0.5 + rng.random(K)) * flip(K)).astype(np.float16)
    g = (0.1 * (0.5 + rng.random(N)) * flip(N)).astype(np.float16)
    b = (0.1 * rng.standard_normal(N)).astype(np.float16) if bias else None
    y_ref, u_ref = coracle.forward_f16(packed, x, h, g, b, return_pre_ln=True)
    m = _make_layer(K, N, torch.float16, dev, packed, h, g, b)
    xt = _t(x, dev)
    from onebit_amd import _lib
    T = int(np.prod(lead))
    ws = _lib.load().onebit_linear_workspace_bytes(T, K, N, _lib.ONEBIT_F16)
    assert ws >= T * K * 2 + 2 * T * N * 4                                  # the route's workspace: scaled rows + >= 2 slices of fp32 sums
    y = m(xt).cpu().numpy().astype(np.float32)
    m.layernorm = torch.nn.Identity()
    m.bias = None
    u = m(xt).cpu().numpy().astype(np.float32)
    y_ref, u_ref = np.asarray(y_ref, np.float32), np.asarray(u_ref, np.float32)
    floor = np.abs(u_ref).reshape(-1, N).max(axis=1).reshape(*lead, 1) * 2.0 ** -9
    bad = np.abs(u - u_ref) > 2.001 * np.maximum(np.abs(u_ref), floor) * FP16_ULP
    assert not bad.any(), ("pre-LN beyond 2 fp16 ulps (of max(|u|, 2^-9 row max))", int(bad.sum()))
    assert (u != u_ref).mean() <= 0.02
    rel = np.linalg.norm(y - y_ref) / (np.linalg.norm(y_ref) + 1e-30)
    assert rel <= 1e-3, rel
    assert np.abs(y - y_ref).max() <= 2.5 * FP16_ULP * max(1.0, float(np.abs(y_ref).max()))


def test_forward_random_mid_size_shapes_vs_oracle(dev, coracle):
    """Seeded random (T, K, N) in the band where round 6 changed the routing of onebit_linear_forward (65 .. 641 rows; K-sliced GEMM with
    2 .. 4 slices, the LDS-DMA GEMM from its tile threshold on, ragged token / row tiles, N % 256 != 0, bias) against the oracle; bars as in
    test_forward_fp16_k_sliced_route_vs_oracle."""
    rng = np.random.default_rng(20260101)
    for case in range(7):               # (the oracle's time bounds the case count: ~8 s each)
        K = int(rng.choice([2048, 2816, 4096, 5632, 1024, 3072]))
        N = int(rng.integers(16, 300)) * 8
        T = int(rng.choice([65, 70, 128, 129, 191, 192, 200, 257, 320, 384, 500, 641]))
        bias = bool(rng.integers(0, 2))
        packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
        x = rng.standard_normal((T, K)).astype(np.float16)
        flip = lambda n: np.where(rng.random(n) < 0.1, -1.0, 1.0)
        h = (0.1 * (0.5 + rng.random(K)) * flip(K)).astype(np.float16)
        g = (0.1 * (0.5 + rng.random(N)) * flip(N)).astype(np.float16)
        b = (0.1 * rng.standard_normal(N)).astype(np.float16) if bias else None
        y_ref, u_ref = coracle.forward_f16(packed, x, h, g, b, return_pre_ln=True)
        m = _make_layer(K, N, torch.float16, dev, packed, h, g, b)
        xt = _t(x, dev)
        y = m(xt).cpu().numpy().astype(np.float32)
        m.layernorm = torch.nn.Identity()
        m.bias = None
        u = m(xt).cpu().numpy().astype(np.float32)
        y_ref, u_ref = np.asarray(y_ref, np.float32), np.asarray(u_ref, np.float32)
        tag = (case, T, K, N, bias)
        floor = np.abs(u_ref).max(axis=1, keepdims=True) * 2.0 ** -9
        bad = np.abs(u - u_ref) > 2.001 * np.maximum(np.abs(u_ref), floor) * FP16_ULP
        assert not bad.any(), (tag, int(bad.sum()))
        assert (u != u_ref).mean() <= 0.02, tag
        rel = np.linalg.norm(y - y_ref) / (np.linalg.norm(y_ref) + 1e-30)
        assert rel <= 1e-3, (tag, rel)
        assert np.abs(y - y_ref).max() <= 2.5 * FP16_ULP * max(1.0, float(np.abs(y_ref).max())), tag


@pytest.mark.parametrize("lead,K,N,bias", [((2,), 256, 80, True), ((1,), 4096, 512, False), ((3,), 40, 9, False)])
def test_forward_fp32_vs_oracle(dev, coracle, lead, K, N, bias):
    rng = np.random.default_rng(K + N)
    packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
    x = rng.standard_normal((*lead, K)).astype(np.float32)
    h = (0.5 + rng.random(K)).astype(np.float32)
    g = (0.5 + rng.random(N)).astype(np.float32)
    b = (0.1 * rng.standard_normal(N)).astype(np.float32) if bias else None
    y_ref = coracle.forward_f32(packed, x, h, g, b)
    m = _make_layer(K, N, torch.float32, dev, packed, h, g, b)
    y = m(_t(x, dev))
    assert y.dtype == torch.float32
    assert np.abs(y.cpu().numpy() - y_ref).max() <= 2e-4
    # fp16 input with fp32 parameters promotes to fp32, as in the reference
    y2 = m(_t(x, dev).half())
    assert y2.dtype == torch.float32


def test_empty_and_errors(dev):
    from onebit_amd import BitLinearInf
    m = BitLinearInf(64, 48, dtype=torch.float16).to(dev)
    y = m(torch.zeros(0, 64, dtype=torch.float16, device=dev))
    assert y.shape == (0, 48)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 64, dtype=torch.float32, device=dev))     # fp32 x with fp16 params
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 32, dtype=torch.float16, device=dev))
    # reset state: W = +1, g = h = 1 -> constant rows -> LayerNorm output ~ 0
    x = torch.randn(3, 64, device=dev).half()
    assert m(x).float().abs().max() <= 1e-3


def test_pack_unpack_bit_exact(dev, coracle, golden_dir):
    from onebit_amd import fp16_to_int8, int8_to_fp16, pack_signs
    z = np.load(os.path.join(golden_dir, "pack.npz"))
    for i in range(int(z["n_cases"])):
        s = z[f"signs_{i}"]
        for dt in (torch.float32, torch.float16):
            got = fp16_to_int8(_t(s, dev).to(dt)).cpu().numpy()
            np.testing.assert_array_equal(got, z[f"packed_f32_{i}"])
    np.testing.assert_array_equal(pack_signs(_t(z["latent_w"], dev)).cpu().numpy(), z["latent_packed"])
    zu = np.load(os.path.join(golden_dir, "unpack.npz"))
    for dt, name in ((torch.float32, "f32"), (torch.float16, "f16")):
        got = int8_to_fp16(_t(zu["bytes"], dev), dt).float().cpu().numpy()
        np.testing.assert_array_equal(got, zu[f"dense_{name}"])
    # full-size round trip (7B gate shape) + agreement with the oracle packer
    g = torch.Generator(device="cpu").manual_seed(5)
    w = torch.randn(11008, 4096, generator=g)
    w[5, 7] = 0.0
    w[6, 8] = float("nan")
    p = pack_signs(w.to(dev))
    np.testing.assert_array_equal(p.cpu().numpy()[:64], coracle.pack_signs(w[:64].numpy()))
    d = int8_to_fp16(p, torch.float16)
    exp = torch.where(w.to(dev) < 0, -1.0, 1.0).half()
    assert torch.equal(d, exp)
    assert torch.equal(pack_signs(d), p)                           # idempotent
    with pytest.raises(ValueError):
        pack_signs(torch.ones(2, 12, device=dev))


def test_fp16_to_int8_on_non_sign_tensors(dev, coracle, golden_dir):
    """`fp16_to_int8` is the reference function, not only on +-1: convert_llama_to_infer_ckpt.py:10 truncates
    (0 - s + 1) / 2 to uint8 and the uint8 matmul (:12-13) wraps -- s = -0.5 packs as +1, s = -3 sets the NEXT bit.
    Bit-exact against bytes the reference itself produced (pack_nonsign.npz), fp32 and fp16 inputs, and against
    the oracle on a larger random tensor of the same domain (s <= 1)."""
    from onebit_amd import fp16_to_int8, pack_signs
    z = np.load(os.path.join(golden_dir, "pack_nonsign.npz"))
    for i in range(int(z["n_cases"])):
        s = z[f"s_{i}"]
        for dt, name in ((torch.float32, "f32"), (torch.float16, "f16")):
            got = fp16_to_int8(_t(s, dev).to(dt)).cpu().numpy()
            np.testing.assert_array_equal(got, z[f"packed_{name}_{i}"])
    rng = np.random.default_rng(17)
    vals = np.array([-509.0, -64.0, -3.0, -2.5, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0], np.float32)
    s = vals[rng.integers(0, len(vals), (96, 4096))]
    np.testing.assert_array_equal(fp16_to_int8(_t(s, dev)).cpu().numpy(), coracle.fp16_to_int8(s))
    # on sign values the two packers agree (what the converter feeds it, :30)
    sg = np.sign(rng.standard_normal((64, 512))).astype(np.float32)
    assert torch.equal(fp16_to_int8(_t(sg, dev)), pack_signs(_t(sg, dev)))
    with pytest.raises(ValueError):
        fp16_to_int8(torch.ones(2, 12, device=dev))


def test_sign_flip_and_row_permutation_properties(dev):
    """Size-independent properties at the full 4096 -> 11008 shape: complementing every weight bit
    negates the output exactly (LN(-u) = -LN(u), fp16 rounding is sign-symmetric); permuting weight
    rows (with g) permutes outputs exactly."""
    from onebit_amd import BitLinearInf
    K, N = 4096, 11008
    g = torch.Generator(device="cpu").manual_seed(11)
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(2, K, generator=g).half().to(dev)
    y = m(x)
    m2 = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m2.load_state_dict(m.state_dict())
    m2.weight.data = ~m.weight.data
    assert torch.equal(m2(x), -y)
    perm = torch.randperm(N, generator=g).to(dev)
    m2.weight.data = m.weight.data[perm]
    m2.weight_scale.data = m.weight_scale.data[perm]
    # LayerNorm statistics are summed in another order -> allow one fp16 ulp on y; pre-LN is exact
    assert (m2(x).float() - y[:, perm].float()).abs().max() <= 2 * FP16_ULP * max(1.0, float(y.abs().max()))
    m.layernorm = torch.nn.Identity()
    m2.layernorm = torch.nn.Identity()
    assert torch.equal(m2(x), m(x)[:, perm])


def test_k_sharded_partials_match_full(dev, coracle):
    """onebit_matmul_partial over K-slices (aliasing column slices of the packed matrix and of x in
    place) + onebit_scale_layernorm == the single-call forward."""
    from onebit_amd import _lib
    from onebit_amd.bitnet import _stream_ptr
    lib = _lib.load()
    T, K, N, S = 5, 1024, 96, 4
    rng = np.random.default_rng(2)
    packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
    x = rng.standard_normal((T, K)).astype(np.float16)
    h = (0.1 * (0.5 + rng.random(K))).astype(np.float16)
    g = (0.1 * (0.5 + rng.random(N))).astype(np.float16)
    y_ref, u_ref = coracle.forward_f16(packed, x, h, g, None, return_pre_ln=True)
    pt, xt, ht, gt = (_t(a, dev) for a in (packed, x, h, g))
    zsum = torch.zeros(T, N, dtype=torch.float32, device=dev)
    Ks = K // S
    for s in range(S):
        zp = torch.empty(T, N, dtype=torch.float32, device=dev)
        rc = lib.onebit_matmul_partial(pt.data_ptr() + s * Ks // 8, K // 8, xt.data_ptr() + 2 * s * Ks, K,
                                       ht.data_ptr() + 2 * s * Ks, zp.data_ptr(), T, Ks, N, 0, _stream_ptr(dev))
        _lib.check(rc, "matmul_partial")
        zsum += zp
    y = torch.empty(T, N, dtype=torch.float16, device=dev)
    u = torch.empty(T, N, dtype=torch.float16, device=dev)
    rc = lib.onebit_scale_layernorm(zsum.data_ptr(), gt.data_ptr(), None, y.data_ptr(), u.data_ptr(),
                                    T, N, 0, 1e-5, 0, _stream_ptr(dev))
    _lib.check(rc, "scale_layernorm")
    _check_f16(y.cpu().numpy(), u.cpu().numpy(), y_ref, u_ref, "ksharded")


def test_prefill_gemm_matches_single_token_path(dev):
    """Full-width prefill property (BASELINE config 3 shape, fewer tokens): every token of a batched
    call (tiled GEMM kernel) agrees with the same token pushed alone through the T <= 16 kernel."""
    from onebit_amd import BitLinearInf
    K, N, T = 4096, 11008, 300
    g = torch.Generator(device="cpu").manual_seed(21)
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(T, K, generator=g).half().to(dev)
    y = m(x)
    m.layernorm = torch.nn.Identity()
    u = m(x)
    for t in (0, 15, 127, 128, 255, 299):
        u1 = m(x[t:t + 1])
        d = (u[t].float() - u1[0].float()).abs()
        ulp = u1[0].float().abs().clamp_min(2.0 ** -14) * FP16_ULP
        assert (d <= 2.001 * ulp).all(), t
        assert (u[t] != u1[0]).float().mean() <= 0.02, t
    m.layernorm = torch.nn.LayerNorm(N, elementwise_affine=False)
    y1 = m(x[128:129])
    assert (y[128].float() - y1[0].float()).abs().max() <= 2.5 * FP16_ULP * max(1.0, float(y1.abs().max()))


def test_k_sharded_forward_hip_world1(dev):
    """The K-sharded driver with the HIP callbacks over RCCL at world_size 1 (the only size a 1-GPU
    box offers; sizes 2/3 are covered on CPU with gloo in tests/test_sharded_cpu.py)."""
    import os
    import torch.distributed as dist
    from onebit_amd import BitLinearInf
    from onebit_amd.sharded import k_sharded_forward, shard_k
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        K, N, T = 1024, 272, 37
        g = torch.Generator(device="cpu").manual_seed(8)
        m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
        m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
        m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
        m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
        x = torch.randn(T, K, generator=g).half().to(dev)
        ref = m(x)
        for mode in ("rs_ag", "allreduce"):
            # emulate 4 K-shards on one device: sum the partials by hand through the same callbacks
            from onebit_amd.sharded import hip_epilogue, hip_partial
            z = sum(hip_partial(shard_k(m.weight.data, m.input_factor.data, m.weight_scale.data, None, r, 4, copy=(r % 2 == 0)),
                                x[:, r * 256:(r + 1) * 256]) for r in range(4))
            y4 = hip_epilogue(shard_k(m.weight.data, m.input_factor.data, m.weight_scale.data, None, 0, 4), z, torch.float16)
            assert (y4.float() - ref.float()).abs().max() <= 2.5 * FP16_ULP * max(1.0, float(ref.abs().max()))
            y = k_sharded_forward(shard_k(m.weight.data, m.input_factor.data, m.weight_scale.data, None, 0, 1), x, mode=mode)
            assert (y.float() - ref.float()).abs().max() <= 2.5 * FP16_ULP * max(1.0, float(ref.abs().max()))
    finally:
        if created:
            dist.destroy_process_group()


def test_n_sharded_rows_match_full_forward():
    """N-sharding at world size 1 and as two hand-made row slices: every slice's columns, normalised
    with the statistics of the complete row, equal the full forward (host glue in fp32 torch ops)."""
    from onebit_amd import BitLinearInf
    from onebit_amd.sharded import hip_rows_u, n_sharded_forward, shard_n
    dev = torch.device("cuda:0")
    K, N, T = 512, 1408, 9
    g = torch.Generator().manual_seed(3)
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(T, K, generator=g).half().to(dev)
    y = m(x)
    full = shard_n(m.weight.data, m.input_factor.data, m.weight_scale.data, None, 0, 1)
    y1 = n_sharded_forward(full, x)
    assert (y1.float() - y.float()).abs().max() <= 2.5 * FP16_ULP * max(1.0, float(y.abs().max()))
    # two slices: u of each slice is exactly the corresponding columns of the full pre-LN output
    m.layernorm = torch.nn.Identity()
    u = m(x)
    for r in range(2):
        sh = shard_n(m.weight.data, m.input_factor.data, m.weight_scale.data, None, r, 2)
        assert torch.equal(hip_rows_u(sh, x), u[:, sh.n0:sh.n1])


def test_forward_shape_sweep_hits_every_kernel_route(dev, coracle):
    """T x K x N sweep across the dispatch boundaries of onebit_linear_forward (decode GEMV at T = 1,
    skinny kernel 2..64 with 16/32/64-token tiles, 64-row GEMM tiles for unaligned rows, 128 x 128
    prefill tile, ragged last tiles in every dimension) against the oracle, fp16."""
    rng = np.random.default_rng(123)
    shapes = []
    for T in (1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 129):
        for (K, N) in ((512, 64), (1024, 80), (640, 200), (96, 48)):      # 96: K % 128 != 0 -> no skinny / aligned GEMV
            shapes.append((T, K, N))
    shapes += [(5, 2048, 24), (40, 1536, 136), (130, 256, 264), (1, 2560, 72)]
    for (T, K, N) in shapes:
        packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
        h = (0.1 * (0.5 + rng.random(K)) * np.where(rng.random(K) < 0.1, -1, 1)).astype(np.float16)
        g = (0.1 * (0.5 + rng.random(N)) * np.where(rng.random(N) < 0.1, -1, 1)).astype(np.float16)
        x = rng.standard_normal((T, K)).astype(np.float16)
        y_ref, u_ref = coracle.forward_f16(packed, x, h, g, None, return_pre_ln=True)
        m = _make_layer(K, N, torch.float16, dev, packed, h, g)
        y = m(_t(x, dev)).cpu().numpy()
        m.layernorm = torch.nn.Identity()
        u = m(_t(x, dev)).cpu().numpy()
        _check_f16(y, u, y_ref, u_ref, (T, K, N))


@pytest.mark.parametrize("env", [{"OB_GEMM3": "2"}, {"OB_GEMM3": "0", "OB_GEMM2": "2"}, {"OB_GEMM3": "2", "OB_GEMM4": "1"}])
def test_large_tile_kernels_forced_on_ragged_shapes(dev, env):
    """The 256 x 256 prefill kernels (LDS-DMA ob_gemm3 / register-staged ob_gemm2; round 6: ob_gemm4, the LDS-DMA kernel on the
    32x32x16 MFMA, opt-in) are normally chosen only for grids of >= 4 tiles per CU; forced here (their env switches are read once per process, hence the
    child interpreter) onto small shapes with ragged last tiles in T and N, against the oracle."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "forced_route_child.py")
    r = subprocess.run([sys.executable, child], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forced-route ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
