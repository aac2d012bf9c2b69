"""Pin the oracle (numpy + C restatements) against fixtures produced by the
imported reference (tests/golden/gen_goldens.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle as O

FP16_ULP_AT_1 = 2.0 ** -10


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_pack_matches_reference(golden_dir, coracle):
    z = _load(golden_dir, "pack.npz")
    for i in range(int(z["n_cases"])):
        s = z[f"signs_{i}"]
        for dn in ("f32", "f16"):
            ref = z[f"packed_{dn}_{i}"]
            assert ref.dtype == np.int8
            np.testing.assert_array_equal(O.np_fp16_to_int8(s), ref)
            np.testing.assert_array_equal(coracle.fp16_to_int8(s), ref)
            # signs with zeros: sign-then-pack agrees as well
            np.testing.assert_array_equal(coracle.pack_signs(s), ref)
    np.testing.assert_array_equal(O.np_pack_signs(z["latent_w"]), z["latent_packed"])
    np.testing.assert_array_equal(coracle.pack_signs(z["latent_w"]), z["latent_packed"])


def test_fp16_to_int8_non_sign_values(golden_dir, coracle):
    """The reference's packer on tensors that are not +-1 (its asserts are commented out,
    convert_llama_to_infer_ckpt.py:8-9): truncation of (1 - s) / 2 and the uint8 wrap, both oracle statements
    against bytes the reference produced (tests/golden/gen_goldens_pack_nonsign.py)."""
    z = _load(golden_dir, "pack_nonsign.npz")
    for i in range(int(z["n_cases"])):
        s = z[f"s_{i}"]
        for dn in ("f32", "f16"):
            np.testing.assert_array_equal(coracle.fp16_to_int8(s), z[f"packed_{dn}_{i}"])
            np.testing.assert_array_equal(O.np_fp16_to_int8(s), z[f"packed_{dn}_{i}"])


def test_pack_engineered_bytes(golden_dir):
    z = _load(golden_dir, "pack.npz")
    row0 = z["packed_f32_1"][0].view(np.uint8)       # (5, 32) case, row 0
    assert list(row0) == [0x00, 0x01, 0x80, 0xFF]


def test_pack_rejects_bad_k(coracle):
    with pytest.raises(ValueError):
        O.np_fp16_to_int8(np.ones((2, 12), np.float32))
    with pytest.raises(ValueError):
        coracle.pack_signs(np.ones((2, 12), np.float32))


def test_unpack_all_bytes(golden_dir, coracle):
    z = _load(golden_dir, "unpack.npz")
    b = z["bytes"]
    for dn in ("f32", "f16"):
        np.testing.assert_array_equal(O.np_int8_to_fp(b), z[f"dense_{dn}"])
        np.testing.assert_array_equal(coracle.unpack(b), z[f"dense_{dn}"])


def test_pack_unpack_roundtrip(coracle):
    rng = np.random.default_rng(7)
    s = np.where(rng.random((33, 128)) < 0.5, -1.0, 1.0).astype(np.float32)
    np.testing.assert_array_equal(coracle.unpack(coracle.pack_signs(s)), s)
    # int32 view of the packed rows is "32 signs per word, LSB first" 
    p = coracle.pack_signs(s)
    w = np.ascontiguousarray(p).view(np.uint32)
    bits = (w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1
    np.testing.assert_array_equal((1.0 - 2.0 * bits).reshape(33, 128), s)


def _cases(z):
    meta = z["meta_idx_K_N_bias_nlead"]
    for row, dn, sp in zip(meta, z["meta_dtype"], z["meta_special"]):
        idx, K, N, bias, nlead = (int(v) for v in row)
        p = f"c{idx}_"
        yield dict(idx=idx, K=K, N=N, dtype=str(dn), special=str(sp),
                   packed=z[p + "packed"], x=z[p + "x"], h=z[p + "h"], g=z[p + "g"],
                   bias=z[p + "bias"] if bias else None, y=z[p + "y"], u=z[p + "u"])


def test_forward_fp32_matches_reference(golden_dir, coracle):
    z = _load(golden_dir, "forward.npz")
    n = 0
    for c in _cases(z):
        if c["dtype"] != "f32":
            continue
        for impl in (coracle.forward_f32, O.np_forward_f32):
            y, u = impl(c["packed"], c["x"], c["h"], c["g"], c["bias"], return_pre_ln=True)
            assert y.shape == c["y"].shape and y.dtype == np.float32
            # reference accumulates in fp32, oracle in fp64: round-off only
            scale_u = np.abs(c["u"]).max() + 1e-30
            assert np.abs(u - c["u"]).max() <= 2e-5 * scale_u, c["idx"]
            if c["special"] != "reset":
                assert np.abs(y - c["y"]).max() <= 2e-4, c["idx"]
        n += 1
    assert n == 16


def test_forward_fp16_matches_reference(golden_dir, coracle):
    z = _load(golden_dir, "forward.npz")
    n = 0
    for c in _cases(z):
        if c["dtype"] != "f16":
            continue
        for impl in (coracle.forward_f16, O.np_forward_f16):
            y, u = impl(c["packed"], c["x"], c["h"], c["g"], c["bias"], return_pre_ln=True)
            assert y.dtype == np.float16 and y.shape == c["y"].shape
            yf, rf = y.astype(np.float32), c["y"].astype(np.float32)
            uf, ruf = u.astype(np.float32), c["u"].astype(np.float32)
            # pre-LN: same rounding points; the reference's CPU fp16 GEMM may
            # accumulate in another order -> rare 1-ulp flips
            ulp_u = np.maximum(np.abs(ruf), 2.0 ** -14) * FP16_ULP_AT_1
            assert (np.abs(uf - ruf) <= 2.0 * ulp_u).all(), c["idx"]
            if c["special"] == "reset":
                continue
            rel_l2 = np.linalg.norm(yf - rf) / (np.linalg.norm(rf) + 1e-30)
            assert rel_l2 <= 1e-3, (c["idx"], rel_l2)
            assert np.abs(yf - rf).max() <= 4e-3 * max(1.0, np.abs(rf).max()), c["idx"]
        n += 1
    assert n == 16


def test_forward_reset_state_constant_rows(golden_dir, coracle):
    """W=+1, g=h=1: every output of a token is the same number, variance 0;
    LayerNorm returns 0 * rsqrt(eps) = 0 (fp32) -- pins eps / biased-variance handling."""
    z = _load(golden_dir, "forward.npz")
    for c in _cases(z):
        if c["special"] != "reset":
            continue
        fn = coracle.forward_f32 if c["dtype"] == "f32" else coracle.forward_f16
        y, u = fn(c["packed"], c["x"], c["h"], c["g"], None, return_pre_ln=True)
        assert np.abs(u.astype(np.float32) - c["u"].astype(np.float32)).max() <= 1e-2
        # all u equal -> (u - mean) ~ 0 -> y ~ 0 (reference: 0 in fp32, 2.9e-6 in fp16)
        assert np.abs(y.astype(np.float32) - c["y"].astype(np.float32)).max() <= 1e-3


def test_c_and_numpy_oracles_agree_bitwise_fp16(coracle):
    rng = np.random.default_rng(3)
    for (T, K, N) in [(1, 64, 16), (3, 256, 80), (2, 1376, 96)]:
        packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
        x = rng.standard_normal((T, K)).astype(np.float16)
        h = (0.1 * (0.5 + rng.random(K))).astype(np.float16)
        g = (0.1 * (0.5 + rng.random(N))).astype(np.float16)
        b = (0.1 * rng.standard_normal(N)).astype(np.float16)
        y1, u1 = coracle.forward_f16(packed, x, h, g, b, return_pre_ln=True)
        y2, u2 = O.np_forward_f16(packed, x, h, g, b, return_pre_ln=True)
        np.testing.assert_array_equal(u1.view(np.uint16), u2.view(np.uint16))
        assert (y1.view(np.uint16) != y2.view(np.uint16)).mean() < 0.01
        assert np.abs(y1.astype(np.float32) - y2.astype(np.float32)).max() < 2e-3


def test_half_conversion_exhaustive(coracle):
    allh = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for b in range(0, 65536, 257):
        got = coracle.lib.ob_half_to_float(int(b))
        exp = f[b]
        assert (np.isnan(got) and np.isnan(exp)) or got == exp
        if not np.isnan(exp):
            assert coracle.lib.ob_float_to_half(float(exp)) == b
    rng = np.random.default_rng(0)
    xs = (rng.standard_normal(20000) * np.exp(rng.uniform(-20, 11, 20000))).astype(np.float32)
    got = np.array([coracle.lib.ob_float_to_half(float(v)) for v in xs], dtype=np.uint16)
    np.testing.assert_array_equal(got, xs.astype(np.float16).view(np.uint16))
