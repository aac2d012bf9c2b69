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
