"""ctypes binding of libonebit_hip.so (the C ABI declared in include/onebit.h).

The product path has no CPU fallback: if the shared library is missing or a
symbol is absent this module raises, loudly.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# ONEBIT_LIB: an alternative build of the SAME sources (A/B builds made by tools/variant_bench.py with -D switches);
# never a different implementation -- the ABI version and every symbol are still checked below
LIB_PATH = os.environ.get("ONEBIT_LIB") or os.path.join(_HERE, "csrc", "libonebit_hip.so")

ONEBIT_F16, ONEBIT_F32 = 0, 1
FLAG_SKIP_LN = 1
FLAG_Q_TOKEN_MAJOR = 2      # onebit_rows_qkv_rope
FLAG_PRESCALED = 4
FLAG_TILE_STATS = 8
ABI_VERSION = 9

# name -> (restype, argtypes); must list every symbol include/onebit.h declares
_i64, _vp, _int, _f, _u = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
SYMBOLS = {
    "onebit_abi_version": (_int, []),
    "onebit_last_error": (ctypes.c_char_p, []),
    "onebit_pack_signs": (_int, [_vp, _int, _vp, _i64, _i64, _vp]),
    "onebit_fp16_to_int8": (_int, [_vp, _int, _vp, _i64, _i64, _vp]),
    "onebit_unpack_signs": (_int, [_vp, _vp, _int, _i64, _i64, _vp]),
    "onebit_linear_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64, _int]),
    "onebit_linear_forward": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t,
                                     _i64, _i64, _i64, _int, _f, _u, _vp]),
    "onebit_matmul_partial": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "onebit_matmul_partial_ws": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, ctypes.c_size_t, _i64, _i64, _i64, _int, _vp]),
    "onebit_scale_layernorm": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f, _u, _vp]),
    "onebit_row_stats": (_int, [_vp, _vp, _i64, _i64, _int, _vp]),
    "onebit_normalize_rows": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "onebit_linear_tile_stats_ok": (_int, [_i64, _i64, _i64, _int]),
    "onebit_tile_stats_combine": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "onebit_rows_res_ln_rms": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _i64, _i64, _f, _f, _vp]),
    "onebit_rows_swiglu": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f, _vp]),
    "onebit_linear_prescaled_ok": (_int, [_i64, _i64, _i64, _int]),
    "onebit_rows_swiglu_stats": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f, _vp]),
    "onebit_rows_qkv_rope_stats": (_int, [_vp] * 9 + [_i64, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i64, _i64, _i64, ctypes.c_float, ctypes.c_uint, _vp]),
    "onebit_rows_qkv_rope": (_int, [_vp] * 8 + [_i64, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i64, _i64, _i64, ctypes.c_float, ctypes.c_uint, _vp]),
    "onebit_attention_prefill": (_int, [_vp] * 5 + [_i64, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i64, _i64, _vp]),
    "onebit_linear_group_prescaled": (_int, [_vp, _vp, _vp, ctypes.c_int32, _i64, _vp]),
    "onebit_rows_res_ln_rms_bias": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _i64, _i64, _f, _f, _vp]),
    "onebit_rows_qkv_rope_ragged": (_int, [_vp] * 13 + [_i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i64, _i64, _i64, ctypes.c_float, _vp]),
    "onebit_attention_ragged": (_int, [_vp] * 6 + [ctypes.c_int32] * 4 + [_i64, _i64, _vp]),
    "onebit_attention_decode_scratch_bytes": (ctypes.c_size_t, [_i64, ctypes.c_int32, ctypes.c_int32]),
    "onebit_attention_decode_rows": (_int, [_vp] * 7 + [_i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i64, _i64, ctypes.c_int32, ctypes.c_int32,
                                                      _vp, ctypes.c_size_t, _vp]),
    "onebit_attention_decode_rows_fused": (_int, [_vp] * 17 + [_i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i64, _i64, _i64, ctypes.c_int32,
                                                            ctypes.c_int32, _f, _vp, ctypes.c_size_t, _vp]),
    "onebit_mixed_workspace_bytes": (ctypes.c_size_t, [_vp, _i64, ctypes.c_int32, ctypes.c_int32]),
    "onebit_mixed_step": (_int, [_vp, _vp, _vp]),     # (onebit_model_t*, onebit_mixed_state_t*, stream)
    "onebit_attn_scratch_bytes": (ctypes.c_size_t, [_vp, _int]),
    "onebit_decode_stats_floats": (ctypes.c_size_t, [_vp]),
    "onebit_batch_stats_floats": (ctypes.c_size_t, [_vp, ctypes.c_int32]),
    "onebit_decode_step": (_int, [_vp, _vp, _vp]),      # (onebit_model_t*, onebit_decode_state_t*, stream)
    "onebit_decode_step_batched": (_int, [_vp, _vp, _vp]),   # (onebit_model_t*, onebit_batch_state_t*, stream)
    "onebit_decode_step_ksharded": (_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp]),   # (model*, onebit_kshard_state_t*, layer, segment, stream)
    "onebit_debug_fill_lds": (_int, [ctypes.c_uint32, _vp]),
    "onebit_fused_gemv": (_int, [_vp, _vp, _int, _int, _vp, _vp]),
    "onebit_train_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64, _int]),
    "onebit_train_forward": (_int, [_vp] * 8 + [_i64, _i64, _i64, _int, _f, _vp]),
    "onebit_train_backward": (_int, [_vp] * 13 + [ctypes.c_size_t, _i64, _i64, _i64, _int, _vp]),
}

_lock = threading.Lock()
_lib = None


class OneBitLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and type the shared library.  Raises if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise OneBitLibraryError(
                f"{LIB_PATH} not found: build it with `python -m onebit_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise OneBitLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise OneBitLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        v = lib.onebit_abi_version()
        if v != ABI_VERSION:
            raise OneBitLibraryError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc: int, what: str):
    """Map a C-ABI return code to a Python exception (argument errors -> ValueError,
    like the reference's view()/shape errors; HIP errors -> RuntimeError)."""
    if rc == 0:
        return
    msg = load().onebit_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise RuntimeError(f"{what}: HIP error {rc}: {msg}")
