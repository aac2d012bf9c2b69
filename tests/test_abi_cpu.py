"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, rejects bad arguments without touching a GPU, and the nn.Module mirrors the
reference's constructor / parameter contract."""
import ctypes
import os
import re

import pytest
import torch

from onebit_amd import _lib
from onebit_amd.bitnet import BitLinearInf, OneBitLinear

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "onebit.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(onebit_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _header_symbols()
    assert syms, "no declarations found in include/onebit.h"
    assert syms == set(_lib.SYMBOLS), "ctypes binding and header disagree"
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.onebit_abi_version() == _lib.ABI_VERSION


def test_argument_errors_without_gpu():
    lib = _lib.load()
    E_ARG, E_SHAPE, E_ALIGN, E_DTYPE, E_FLAG = -1, -2, -3, -4, -6
    assert lib.onebit_pack_signs(None, 0, None, 2, 12, None) == E_SHAPE
    assert b"multiple of 8" in lib.onebit_last_error()
    assert lib.onebit_pack_signs(None, 7, None, 2, 16, None) == E_DTYPE
    assert lib.onebit_pack_signs(None, 0, None, 2, 16, None) == E_ARG
    assert lib.onebit_pack_signs(None, 0, None, 0, 16, None) == 0          # empty is fine
    assert lib.onebit_unpack_signs(None, None, 0, -1, 16, None) == E_ARG
    buf = ctypes.create_string_buffer(4096)
    a = (ctypes.addressof(buf) + 63) & ~63
    fwd = lambda **kw: lib.onebit_linear_forward(
        kw.get("p", a), kw.get("ldw", 8), kw.get("x", a), a, a, None, kw.get("y", a), None, None, 0,
        kw.get("T", 1), kw.get("K", 64), kw.get("N", 4), kw.get("dt", 0), 1e-5, kw.get("fl", 0), None)
    assert fwd(K=44) == E_SHAPE
    assert fwd(dt=3) == E_DTYPE
    assert fwd(p=None) == E_ARG
    assert fwd(x=a + 2) == E_ALIGN
    assert fwd(ldw=6) == E_ALIGN          # pitch smaller than K/8
    assert fwd(K=40, ldw=5) == -5         # generic shape without workspace
    assert fwd(fl=64) == E_FLAG
    assert fwd(T=0) == 0
    # N-sharded LayerNorm halves and the batched step reject bad arguments before any launch
    assert lib.onebit_row_stats(a, a, 2, 0, 0, None) == E_SHAPE
    assert lib.onebit_row_stats(None, a, 2, 8, 0, None) == E_ARG
    assert lib.onebit_row_stats(a, a, 0, 8, 0, None) == 0
    assert lib.onebit_row_stats(a, a, 2, 8, 5, None) == E_DTYPE
    assert lib.onebit_normalize_rows(a, None, a, None, a, 2, 8, 0, None) == E_ARG
    assert lib.onebit_normalize_rows(a, a, a, None, a, -1, 8, 0, None) == E_ARG
    assert lib.onebit_decode_step_batched(None, None, None) == E_ARG
    from onebit_amd.engine import _BatchState, _Layer, _Model
    layers = (_Layer * 1)()
    m = _Model(1, 64, 128, 2, 2, 32, 16, 8, 1e-6, 1e-5, layers, a, a, a, a, a)
    st = _BatchState(ctypes.sizeof(_BatchState), 1, a, a, a, a, a, a, a, a, a, a, a, a, a, a)
    lib.onebit_decode_step_batched.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_BatchState), ctypes.c_void_p]
    assert lib.onebit_decode_step_batched(ctypes.byref(m), ctypes.byref(st), None) == E_SHAPE      # batch 1
    assert b"batch" in lib.onebit_last_error()
    # ABI 8: a state built against another header (another size) is refused before any field is read
    st.struct_size = ctypes.sizeof(_BatchState) - 8
    assert lib.onebit_decode_step_batched(ctypes.byref(m), ctypes.byref(st), None) == E_ARG
    assert b"struct_size" in lib.onebit_last_error()
    from onebit_amd.engine import _State
    ds = _State()
    lib.onebit_decode_step.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_State), ctypes.c_void_p]
    assert lib.onebit_decode_step(ctypes.byref(m), ctypes.byref(ds), None) == E_ARG                # struct_size 0
    assert b"struct_size" in lib.onebit_last_error()
    # the K-sharded decode step validates before any launch as well: null / foreign-size state, unknown segment, layer range
    from onebit_amd.engine import _KState
    lib.onebit_decode_step_ksharded.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_KState), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    assert lib.onebit_decode_step_ksharded(None, None, 0, 0, None) == E_ARG
    ks = _KState()
    assert lib.onebit_decode_step_ksharded(ctypes.byref(m), ctypes.byref(ks), 0, 0, None) == E_ARG and b"struct_size" in lib.onebit_last_error()
    ks.struct_size = ctypes.sizeof(_KState)
    m16 = _Model(1, 64, 128, 2, 2, 32, 16, 8, 1e-6, 1e-5, layers, a, a, a, a, a)
    assert lib.onebit_decode_step_ksharded(ctypes.byref(m16), ctypes.byref(ks), 0, 9, None) == E_FLAG      # unknown segment
    assert lib.onebit_decode_step_ksharded(ctypes.byref(m16), ctypes.byref(ks), 5, 0, None) == E_ARG       # layer outside the model
    assert lib.onebit_decode_step_ksharded(ctypes.byref(m16), ctypes.byref(ks), 0, 0, None) == E_ARG       # null buffers
    with pytest.raises(ValueError):
        _lib.check(E_SHAPE, "x")
    with pytest.raises(RuntimeError):
        _lib.check(1, "x")


def test_round6_entry_points_reject_bad_arguments_without_gpu():
    """ABI 9 (ragged rows, key-block decode attention, grouped projections, the mixed step): argument errors come back before any launch."""
    lib = _lib.load()
    E_ARG, E_SHAPE, E_ALIGN, E_WSPACE = -1, -2, -3, -5
    buf = ctypes.create_string_buffer(8192)
    a = (ctypes.addressof(buf) + 255) & ~255
    from onebit_amd.engine import _Layer, _MixedState, _Model, _Proj, _Seg
    # ragged rope: head_dim must be a power of two >= 16; cache rows beyond the rope tables; biases all or none; empty is fine
    rope = lambda **kw: lib.onebit_rows_qkv_rope_ragged(a, a, a, a, a, None, kw.get("pos", a), a, a, a, kw.get("bq", None), None, None, kw.get("T", 2), 2, 2,
                                                        kw.get("D", 64), 4, kw.get("max_len", 16), kw.get("max_pos", 16), 1e-5, None)
    assert rope(D=48) == E_SHAPE and rope(max_len=32) == E_SHAPE and rope(bq=a) == E_ARG and rope(pos=None) == E_ARG and rope(T=0) == 0
    # ragged attention: head_dim 64 / 128, segments inside the cache, no segments = nothing to do
    seg = (_Seg * 2)(_Seg(0, 5, 1, 3), _Seg(5, 4, 9, 0))
    att = lambda **kw: lib.onebit_attention_ragged(a, a, a, a, None, ctypes.cast(seg, ctypes.c_void_p), kw.get("n", 2), 4, kw.get("hkv", 4), kw.get("D", 64),
                                                   kw.get("slots", 4), 16, None)
    assert att(D=32) == E_SHAPE and att(hkv=3) == E_SHAPE and att(n=0) == 0
    assert att() == E_SHAPE and b"segment 1" in lib.onebit_last_error()             # slot 9 of 4
    # key-block decode attention: chunk a multiple of 64, scratch for more than one split
    assert lib.onebit_attention_decode_scratch_bytes(3, 8, 1) == 0
    assert lib.onebit_attention_decode_scratch_bytes(3, 8, 4) >= 3 * 8 * 4 * (128 + 2) * 4 + 3 * 8 * 4
    dec = lambda **kw: lib.onebit_attention_decode_rows(a, a, a, a, None, None, a, 2, 4, 4, kw.get("D", 64), 4, 16, kw.get("chunk", 64), kw.get("ns", 1),
                                                        None, 0, None) if True else None
    assert dec(chunk=96) == E_SHAPE and dec(D=12) == E_SHAPE and dec(ns=3) == E_WSPACE
    fus = lib.onebit_attention_decode_rows_fused(a, a, a, a, a, a, None, None, None, a, a, a, a, a, None, None, a, 2, 4, 4, 48, 4, 16, 16, 64, 1, 1e-5, None, 0, None)
    assert fus == E_SHAPE                                                          # head_dim 48: the rotate_half pairing needs a power of two
    # grouped projections: 1..3 projections, one projection beyond 64 rows goes through onebit_linear_forward
    pr = (_Proj * 3)()
    up = (ctypes.c_void_p * 3)(a, a, a)
    assert lib.onebit_linear_group_prescaled(ctypes.cast(pr, ctypes.c_void_p), up, up, 0, 8, None) == E_ARG
    assert lib.onebit_linear_group_prescaled(ctypes.cast(pr, ctypes.c_void_p), up, up, 1, 500, None) == E_SHAPE
    assert lib.onebit_linear_group_prescaled(ctypes.cast(pr, ctypes.c_void_p), up, up, 2, 0, None) == 0
    # the mixed step: struct size, head_dim, row accounting, workspace
    lib.onebit_mixed_step.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_MixedState), ctypes.c_void_p]
    lib.onebit_mixed_workspace_bytes.argtypes = [ctypes.POINTER(_Model), ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    layers = (_Layer * 1)()
    m = _Model(1, 128, 352, 2, 2, 64, 96, 32, 1e-6, 1e-5, layers, a, a, a, a, a)
    assert lib.onebit_mixed_workspace_bytes(ctypes.byref(m), 0, 4, 0) == 0
    nb = lib.onebit_mixed_workspace_bytes(ctypes.byref(m), 100, 4, 0)
    assert nb > 100 * 128 * 2 * 10 and lib.onebit_mixed_workspace_bytes(ctypes.byref(m), 200, 4, 0) > nb
    st = _MixedState()
    assert lib.onebit_mixed_step(ctypes.byref(m), ctypes.byref(st), None) == E_ARG and b"struct_size" in lib.onebit_last_error()
    sg = (_Seg * 1)(_Seg(1, 5, 0, 0))
    st = _MixedState(ctypes.sizeof(_MixedState), 6, 1, 1, 1, 4, 0, 0, a, a, a, sg, a, a, None, a, a, a, 1 << 30)
    m48 = _Model(1, 96, 352, 2, 2, 48, 96, 32, 1e-6, 1e-5, layers, a, a, a, a, a)
    assert lib.onebit_mixed_step(ctypes.byref(m48), ctypes.byref(st), None) == E_SHAPE        # head_dim 48
    st.n_rows = 9
    assert lib.onebit_mixed_step(ctypes.byref(m), ctypes.byref(st), None) == E_SHAPE and b"n_rows" in lib.onebit_last_error()
    st.n_rows, st.workspace_bytes = 6, 1024
    assert lib.onebit_mixed_step(ctypes.byref(m), ctypes.byref(st), None) == E_WSPACE
    st.n_rows, st.n_dec, st.n_seg = 0, 0, 0
    assert lib.onebit_mixed_step(ctypes.byref(m), ctypes.byref(st), None) == 0                # an empty step


def test_module_contract_matches_reference():
    m = BitLinearInf(64, 48, bias=True, dtype=torch.float16)
    assert OneBitLinear is BitLinearInf
    assert (m.in_features, m.out_features, m.groups) == (64, 48, 1)
    sd = m.state_dict()
    assert list(sd) == ["weight", "weight_scale", "input_factor", "bias"]   # layernorm adds no keys
    assert sd["weight"].dtype == torch.int8 and sd["weight"].shape == (48, 8)
    assert sd["weight_scale"].shape == (48,) and sd["input_factor"].shape == (64,)
    assert all(not p.requires_grad for p in m.parameters())
    assert (m.weight == 0).all() and (m.weight_scale == 1).all() and (m.input_factor == 1).all()
    assert m.bias.abs().max() <= 1 / (8 ** 0.5)                 # fan_in is K // 8, as in the reference
    assert isinstance(m.layernorm, torch.nn.LayerNorm) and m.layernorm.eps == 1e-5
    assert not m.layernorm.elementwise_affine
    m2 = BitLinearInf(64, 48)
    assert m2.bias is None and m2.weight_scale.dtype == torch.float32
    assert list(m2.state_dict()) == ["weight", "weight_scale", "input_factor"]
    # .half() casts floating params only (int8 weight untouched), like from_pretrained(torch_dtype=fp16)
    m2.half()
    assert m2.weight.dtype == torch.int8 and m2.weight_scale.dtype == torch.float16
    # converter-style in-place data assignment keeps working
    m2.weight.data = torch.randint(-128, 127, (48, 8), dtype=torch.int8)
    m2.load_state_dict(m2.state_dict())


def test_cpu_tensors_fail_loudly():
    m = BitLinearInf(64, 48, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 64, dtype=torch.float16))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 32, dtype=torch.float16))
