"""CPU restatement of the reference 1-bit linear layer -- the parity oracle.

TEST INFRASTRUCTURE ONLY.  Nothing under ``onebit_amd/`` imports this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and only as the checker / the CPU timing baseline.

Two independent statements of the same algorithm live here:

* numpy functions (``np_*``) -- vectorised, written straight from the reference
  source lines cited in each docstring (paths relative to ``/root/reference``);
* ``COracle`` -- ctypes wrapper around ``oracle/onebit_oracle.c`` (plain C99).

Parity pinning: the reference ships no tests or golden vectors, so both
statements are pinned by ``tests/golden/*.npz``, generated in the build
container by ``tests/golden/gen_goldens.py`` from the imported reference
module (``transformers/src/transformers/models/bitnet.py``) and its converter's
``fp16_to_int8``.  ``tests/test_oracle_golden.py`` performs the check.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libonebit_oracle.so")


# ----------------------------------------------------------------- numpy ---

def np_fp16_to_int8(signs: np.ndarray) -> np.ndarray:
    """``fp16_to_int8`` -- scripts/convert_llama_to_infer_ckpt.py:7-15.

    ``signs`` is a float array [N, K] holding +1 / -1 (or 0 from torch.sign).
    bit = uint8((0 - s + 1) / 2) (:10; 0 -> 0.5 -> 0, i.e. +1), 8 bits per
    byte LSB-first through a uint8 matmul with [1,2,...,128] (:11-13), result
    reinterpreted as int8 (:13).
    """
    s = np.asarray(signs, dtype=np.float32)
    if s.ndim != 2 or s.shape[1] % 8 != 0:
        raise ValueError("expected [N, K] with K % 8 == 0")
    v = (0.0 - s + 1.0) / 2.0
    bits = np.where(v >= 1.0, np.trunc(v), 0.0).astype(np.uint8)   # .to(uint8) truncates
    r = bits.reshape(s.shape[0], -1, 8).astype(np.uint32)
    mult = np.array([1, 2, 4, 8, 16, 32, 64, 128], dtype=np.uint32)
    packed = ((r * mult).sum(-1) & 0xFF).astype(np.uint8)
    return packed.view(np.int8)


def np_pack_signs(w: np.ndarray) -> np.ndarray:
    """Converter loop body, convert_llama_to_infer_ckpt.py:29-32:
    ``fp16_to_int8(torch.sign(w))``; sign(0) = 0 packs as bit 0 = +1."""
    return np_fp16_to_int8(np.sign(np.asarray(w, dtype=np.float32)))


def np_int8_to_fp(packed: np.ndarray, dtype=np.float32) -> np.ndarray:
    """``BitLinearInf.int8_to_fp16`` -- bitnet.py:98-110.

    bit_k = (byte >> (k % 8)) & 1 (arithmetic shift on int8, :105), value =
    -2*bit + 1 (:109); returns dense [N, 8 * bytes_per_row]."""
    p = np.asarray(packed).view(np.int8)
    shifts = np.arange(8).reshape(1, 1, 8)
    bits = (p[..., None].astype(np.int64) >> shifts) & 1
    return (-2 * bits + 1).reshape(p.shape[0], -1).astype(dtype)


def np_forward_f32(packed, x, h, g, bias=None, eps=1e-5, return_pre_ln=False):
    """``BitLinearInf.forward`` -- bitnet.py:112-122, fp32 parameters.

    Accumulates in fp64, so it is the "exact" evaluation of the reference
    formula; compare fp32 results against it with a small tolerance."""
    W = np_int8_to_fp(packed, np.float64)
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    a = (x.reshape(-1, x.shape[-1]) * np.asarray(h, np.float32)[None, :]).astype(np.float32)
    z = (a.astype(np.float64) @ W.T).astype(np.float32)
    u = (z * np.asarray(g, np.float32)[None, :]).astype(np.float32)
    ud = u.astype(np.float64)
    mean = ud.mean(-1, keepdims=True)
    var = ((ud - mean) ** 2).mean(-1, keepdims=True)
    y = ((ud - mean) / np.sqrt(var + eps)).astype(np.float32)
    if bias is not None:
        y = y + np.asarray(bias, np.float32)[None, :]
    y = y.reshape(*lead, -1)
    if return_pre_ln:
        return y, u.reshape(*lead, -1)
    return y


def np_forward_f16(packed, x, h, g, bias=None, eps=1e-5, return_pre_ln=False):
    """``BitLinearInf.forward`` with fp16 parameters/input, restating the fp16
    rounding points of bitnet.py:113 (x*h), :115 (GEMM output), :116 (in-place
    *g), :118 (LayerNorm: fp32 statistics, fp16 result), :119-120 (bias add)."""
    W = np_int8_to_fp(packed, np.float64)
    x = np.asarray(x, dtype=np.float16)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    a = (x2.astype(np.float32) * np.asarray(h, np.float16).astype(np.float32)[None, :]).astype(np.float16)
    z = (a.astype(np.float64) @ W.T).astype(np.float32).astype(np.float16)
    u = (z.astype(np.float32) * np.asarray(g, np.float16).astype(np.float32)[None, :]).astype(np.float16)
    ud = u.astype(np.float64)
    mean = ud.mean(-1, keepdims=True)
    var = ((ud - mean) ** 2).mean(-1, keepdims=True)
    y = ((ud - mean) / np.sqrt(var + eps)).astype(np.float32).astype(np.float16)
    if bias is not None:
        y = (y.astype(np.float32) + np.asarray(bias, np.float16).astype(np.float32)[None, :]).astype(np.float16)
    y = y.reshape(*lead, -1)
    if return_pre_ln:
        return y, u.reshape(*lead, -1)
    return y


# --------------------------------------------------------------------- C ---

def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/onebit_oracle.c with gcc (seconds). Returns the .so path."""
    src = os.path.join(_HERE, "onebit_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libonebit_oracle.so"])
    return _LIB_PATH


class COracle:
    """ctypes binding of oracle/onebit_oracle.c."""

    def __init__(self):
        self.lib = ctypes.CDLL(build_c_oracle())
        L = self.lib
        i64, p, f = ctypes.c_int64, ctypes.c_void_p, ctypes.c_float
        L.ob_oracle_fp16_to_int8.argtypes = [p, p, i64, i64]
        L.ob_oracle_pack_signs.argtypes = [p, p, i64, i64]
        L.ob_oracle_unpack.argtypes = [p, p, i64, i64]
        L.ob_oracle_unpack.restype = None
        L.ob_oracle_forward_f32.argtypes = [p, p, p, p, p, p, p, i64, i64, i64, f]
        L.ob_oracle_forward_f16.argtypes = [p, p, p, p, p, p, p, i64, i64, i64, f]
        L.ob_oracle_forward_f32_unpack_every_call.argtypes = [p, p, p, p, p, p, i64, i64, i64, f]
        L.ob_oracle_forward_f32_unpack_every_call_mt.argtypes = [p, p, p, p, p, p, i64, i64, i64, f, ctypes.c_int]
        L.ob_half_to_float.argtypes = [ctypes.c_uint16]
        L.ob_half_to_float.restype = ctypes.c_float
        L.ob_float_to_half.argtypes = [ctypes.c_float]
        L.ob_float_to_half.restype = ctypes.c_uint16

    @staticmethod
    def _ptr(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    def fp16_to_int8(self, signs):
        s = np.ascontiguousarray(signs, dtype=np.float32)
        N, K = s.shape
        out = np.empty((N, K // 8), dtype=np.int8)
        rc = self.lib.ob_oracle_fp16_to_int8(self._ptr(s), self._ptr(out), N, K)
        if rc:
            raise ValueError(f"ob_oracle_fp16_to_int8 rc={rc}")
        return out

    def pack_signs(self, w):
        w = np.ascontiguousarray(w, dtype=np.float32)
        N, K = w.shape
        if K % 8:
            raise ValueError("K % 8 != 0")
        out = np.empty((N, K // 8), dtype=np.int8)
        rc = self.lib.ob_oracle_pack_signs(self._ptr(w), self._ptr(out), N, K)
        if rc:
            raise ValueError(f"ob_oracle_pack_signs rc={rc}")
        return out

    def unpack(self, packed):
        p = np.ascontiguousarray(packed).view(np.int8)
        N, KB = p.shape
        out = np.empty((N, KB * 8), dtype=np.float32)
        self.lib.ob_oracle_unpack(self._ptr(p), self._ptr(out), N, KB * 8)
        return out

    def forward_f32(self, packed, x, h, g, bias=None, eps=1e-5, return_pre_ln=False):
        p = np.ascontiguousarray(packed).view(np.int8)
        x = np.ascontiguousarray(x, dtype=np.float32)
        lead, K = x.shape[:-1], x.shape[-1]
        N = p.shape[0]
        T = int(np.prod(lead)) if lead else 1
        h = np.ascontiguousarray(h, np.float32); g = np.ascontiguousarray(g, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        y = np.empty((T, N), np.float32); u = np.empty((T, N), np.float32)
        rc = self.lib.ob_oracle_forward_f32(self._ptr(p), self._ptr(x), self._ptr(h), self._ptr(g),
                                            self._ptr(b), self._ptr(y), self._ptr(u), T, K, N, eps)
        if rc:
            raise ValueError(f"ob_oracle_forward_f32 rc={rc}")
        y = y.reshape(*lead, N)
        return (y, u.reshape(*lead, N)) if return_pre_ln else y

    def forward_f16(self, packed, x, h, g, bias=None, eps=1e-5, return_pre_ln=False):
        p = np.ascontiguousarray(packed).view(np.int8)
        x = np.ascontiguousarray(x, dtype=np.float16)
        lead, K = x.shape[:-1], x.shape[-1]
        N = p.shape[0]
        T = int(np.prod(lead)) if lead else 1
        h = np.ascontiguousarray(h, np.float16); g = np.ascontiguousarray(g, np.float16)
        b = None if bias is None else np.ascontiguousarray(bias, np.float16)
        y = np.empty((T, N), np.float16); u = np.empty((T, N), np.float16)
        rc = self.lib.ob_oracle_forward_f16(self._ptr(p), self._ptr(x), self._ptr(h), self._ptr(g),
                                            self._ptr(b), self._ptr(y), self._ptr(u), T, K, N, eps)
        if rc:
            raise ValueError(f"ob_oracle_forward_f16 rc={rc}")
        y = y.reshape(*lead, N)
        return (y, u.reshape(*lead, N)) if return_pre_ln else y

    def forward_f32_unpack_every_call(self, packed, x, h, g, scratch=None, eps=1e-5, threads=1):
        """Reference-style CPU path (dense +-1 matrix rebuilt on every call); threads > 1: the OpenMP form
        (output rows dealt to the threads, identical results)."""
        p = np.ascontiguousarray(packed).view(np.int8)
        x = np.ascontiguousarray(x, dtype=np.float32)
        lead, K = x.shape[:-1], x.shape[-1]
        N = p.shape[0]
        T = int(np.prod(lead)) if lead else 1
        if scratch is None:
            scratch = np.empty((N, K), np.float32)
        y = np.empty((T, N), np.float32)
        if threads and threads > 1:
            rc = self.lib.ob_oracle_forward_f32_unpack_every_call_mt(
                self._ptr(p), self._ptr(x), self._ptr(np.ascontiguousarray(h, np.float32)),
                self._ptr(np.ascontiguousarray(g, np.float32)), self._ptr(y), self._ptr(scratch), T, K, N, eps, int(threads))
        else:
            rc = self.lib.ob_oracle_forward_f32_unpack_every_call(
                self._ptr(p), self._ptr(x), self._ptr(np.ascontiguousarray(h, np.float32)),
                self._ptr(np.ascontiguousarray(g, np.float32)), self._ptr(y), self._ptr(scratch), T, K, N, eps)
        if rc:
            raise ValueError(f"rc={rc}")
        return y.reshape(*lead, N)


# ------------------------------------------------------ torch-ops CPU port ---
# The reference's CPU path is made of ATen ops (broadcast shift over int64 temporaries, a dense
# matmul, an in-place scale, LayerNorm), and ATen parallelises each of them over the host cores.
# These two functions restate that op sequence (bitnet.py:98-118) with the same torch ops so the
# benchmark can time it with `torch.set_num_threads(os.cpu_count())`, and the
# "unpack once" variant that keeps the dense matrix between calls (what any sane CPU deployment
# would do; it shows how much of the reference's time is re-unpacking).  Test infrastructure /
# bench cpu_baseline only -- never imported by onebit_amd.

def torch_unpack_ref_style(packed, dtype):
    """bitnet.py:98-110 with torch ops: bit_k = (byte >> k) & 1 broadcast over a trailing axis of 8,
    -> dtype, flatten, w = 1 - 2 * bit."""
    import torch
    k = torch.arange(8, device=packed.device).reshape(1, 1, 8)
    bits = torch.bitwise_and(torch.bitwise_right_shift(packed[:, :, None], k), 1).to(dtype)
    return 1 - 2 * bits.reshape(packed.shape[0], packed.shape[1] * 8)


def torch_forward_ref_style(packed, x, h, g, eps=1e-5, dense=None):
    """bitnet.py:112-118 with torch ops; `dense` = a cached result of torch_unpack_ref_style turns the
    call into the unpack-once variant."""
    import torch
    import torch.nn.functional as F
    a = x * h.reshape(1, -1)
    w = torch_unpack_ref_style(packed, g.dtype) if dense is None else dense
    out = F.linear(a, w)
    out = out * g.reshape(1, -1)
    return F.layer_norm(out, (out.shape[-1],), eps=eps)
