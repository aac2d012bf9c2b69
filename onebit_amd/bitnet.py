"""``BitLinearInf`` -- MI355X-native drop-in for OneBit's packed 1-bit linear layer.

Mirrors the reference module ``transformers/src/transformers/models/bitnet.py:71-122``
(class ``BitLinearInf``): same constructor signature, attribute and parameter names,
dtypes, shapes and ``requires_grad=False`` flags, so a state dict produced by the
reference converter (``scripts/convert_llama_to_infer_ckpt.py``) loads unchanged and
the converter's in-place ``module.weight.data = ...`` assignments keep working (the
kernels consume the reference's own ``int8 [N, K/8]`` layout directly; nothing is
cached or re-laid-out).

The forward pass does not unpack the weight matrix (the reference rebuilds a dense
+-1 matrix on every call, bitnet.py:98-110,114): it calls the C ABI of
``libonebit_hip.so`` (``include/onebit.h``), hand-written HIP kernels for gfx950.
There is no CPU fallback; CPU tensors raise.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib

__all__ = ["BitLinearInf", "OneBitLinear", "int8_to_fp16", "fp16_to_int8", "pack_signs"]


def _dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return _lib.ONEBIT_F16
    if dtype == torch.float32:
        return _lib.ONEBIT_F32
    raise TypeError(f"OneBit HIP kernels support float16 and float32 parameters, got {dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device: torch.device) -> int:
    """The current HIP stream of ``device`` as an integer handle (the raw accessor torch's own launchers use: 0.3 us instead of
    the 4 us of building a torch.cuda.Stream object)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on {t.device}; the OneBit path runs only on a ROCm GPU "
            "(hand-written HIP kernels, no CPU fallback)")


def int8_to_fp16(int8_tensor: torch.Tensor, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """Dense +-1 matrix from packed signs -- ``BitLinearInf.int8_to_fp16`` (bitnet.py:98-110)."""
    _require_gpu(int8_tensor, "int8_to_fp16")
    if int8_tensor.dim() != 2 or int8_tensor.dtype not in (torch.int8, torch.uint8):
        raise ValueError("expected a 2-D int8 tensor [N, K/8]")
    p = int8_tensor.contiguous()
    N, KB = p.shape
    out = torch.empty((N, KB * 8), dtype=dtype, device=p.device)
    lib = _lib.load()
    with torch.cuda.device(p.device):
        rc = lib.onebit_unpack_signs(p.data_ptr(), out.data_ptr(), _dtype_code(dtype), N, KB * 8,
                                     _stream_ptr(p.device))
    _lib.check(rc, "onebit_unpack_signs")
    return out


def pack_signs(weight: torch.Tensor) -> torch.Tensor:
    """``fp16_to_int8(torch.sign(weight))`` -- the converter's per-layer step
    (scripts/convert_llama_to_infer_ckpt.py:29-30): int8 [N, K/8], 8 signs per byte
    LSB-first, bit 1 = negative; exact zeros (and NaN) pack as +1."""
    _require_gpu(weight, "pack_signs")
    if weight.dim() != 2:
        raise ValueError("expected a 2-D weight [N, K]")
    N, K = weight.shape
    if K % 8 != 0:
        # the reference's view(N, -1, 8) raises here (convert_llama_to_infer_ckpt.py:11)
        raise ValueError(f"in_features={K} is not a multiple of 8")
    w = weight.contiguous()
    if w.dtype not in (torch.float16, torch.float32):
        w = w.float()
    out = torch.empty((N, K // 8), dtype=torch.int8, device=w.device)
    lib = _lib.load()
    with torch.cuda.device(w.device):
        rc = lib.onebit_pack_signs(w.data_ptr(), _dtype_code(w.dtype), out.data_ptr(), N, K,
                                   _stream_ptr(w.device))
    _lib.check(rc, "onebit_pack_signs")
    return out


def fp16_to_int8(fp16_tensor: torch.Tensor) -> torch.Tensor:
    """The reference's packer under its own name (convert_llama_to_infer_ckpt.py:7-15), with the reference's
    arithmetic on ANY input: ``v = (0 - s + 1) / 2`` in the tensor's dtype, truncated to uint8, bytes LSB-first
    with the uint8 matmul's wrap -- identical to ``pack_signs`` on +1 / -1 / 0 (what ``torch.sign`` produces),
    and equal to the reference elsewhere too (s = -0.5 packs as +1; |s| > 1 spills into the next bit), for
    s <= 1 (``onebit_fp16_to_int8``; fixture tests/golden/pack_nonsign.npz)."""
    _require_gpu(fp16_tensor, "fp16_to_int8")
    if fp16_tensor.dim() != 2:
        raise ValueError("expected a 2-D tensor [N, K]")
    N, K = fp16_tensor.shape
    if K % 8 != 0:
        raise ValueError(f"in_features={K} is not a multiple of 8")     # the reference's view(N, -1, 8) raises (:11)
    s = fp16_tensor.contiguous()
    if s.dtype not in (torch.float16, torch.float32):
        s = s.float()
    out = torch.empty((N, K // 8), dtype=torch.int8, device=s.device)
    lib = _lib.load()
    with torch.cuda.device(s.device):
        rc = lib.onebit_fp16_to_int8(s.data_ptr(), _dtype_code(s.dtype), out.data_ptr(), N, K, _stream_ptr(s.device))
    _lib.check(rc, "onebit_fp16_to_int8")
    return out


class BitLinearInf(nn.Module):
    """``y = LayerNorm(weight_scale * (sign_matrix @ (input_factor * x))) (+ bias)``.

    Parameters (all ``requires_grad=False``, bitnet.py:78-85):
      weight        int8  [out_features, in_features // 8]  packed signs
      weight_scale  dtype [out_features]                    g
      input_factor  dtype [in_features]                     h
      bias          dtype [out_features] or None            added after the LayerNorm
    """

    def __init__(self, in_features, out_features, groups=1, bias=False, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.groups = groups            # accepted and ignored, as in the reference (bitnet.py:77)
        self.weight = nn.Parameter(
            torch.empty((out_features, in_features // 8), device=device, dtype=torch.int8),
            requires_grad=False)
        self.weight_scale = nn.Parameter(torch.empty(out_features, **factory_kwargs), requires_grad=False)
        self.input_factor = nn.Parameter(torch.empty(in_features, **factory_kwargs), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, **factory_kwargs), requires_grad=False)
        else:
            self.register_parameter("bias", None)
        # parameter-free LayerNorm over out_features (bitnet.py:86); kept as a submodule so that
        # `module.layernorm.eps` / replacing it with nn.Identity() behave as in the reference.
        self.layernorm = nn.LayerNorm(out_features, elementwise_affine=False)
        self.reset_parameters()

    def reset_parameters(self):
        # bitnet.py:89-96: g = h = 1, packed weight = 0 (all +1)
        nn.init.constant_(self.weight_scale, 1.0)
        nn.init.constant_(self.input_factor, 1.0)
        with torch.no_grad():
            self.weight.zero_()
        if self.bias is not None:
            fan_in = self.weight.shape[1]           # what _calculate_fan_in_and_fan_out sees: K // 8
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def int8_to_fp16(self, int8_tensor):
        return int8_to_fp16(int8_tensor, self.weight_scale.dtype)

    def extra_repr(self):
        return (f"in_features={self.in_features}, out_features={self.out_features}, "
                f"bias={self.bias is not None}")

    def pre_layernorm(self, input: torch.Tensor) -> torch.Tensor:
        """u = (W . (h * x)) * g, the value the layer's LayerNorm is applied to (bitnet.py:113-116):
        for callers that fuse that LayerNorm with what follows (onebit_rows_* kernels)."""
        if self.bias is not None:
            raise RuntimeError("pre_layernorm: the bias is added after the LayerNorm; use forward")
        return self.forward(input, _pre_ln=True)

    def pre_layernorm_bias_deferred(self, input: torch.Tensor, prescaled: bool = False) -> torch.Tensor:
        """``pre_layernorm`` / ``pre_layernorm_prescaled`` of a layer WITH a bias, for a caller that adds ``self.bias`` itself after
        the LayerNorm it fuses with what follows (onebit_rows_qkv_rope_ragged / onebit_rows_res_ln_rms_bias take the bias)."""
        return self.forward(input, _pre_ln=True, _prescaled=prescaled)

    def prescaled_ok(self, T: int, dtype=torch.float16, bias_deferred: bool = False) -> bool:
        """True when a T-row call of this layer may consume pre-scaled activations
        (``pre_layernorm_prescaled``): it takes the LDS-DMA GEMM, which reads fp16(x * h) rows."""
        prm = self._parameters
        w, g, b = prm.get("weight"), prm.get("weight_scale"), prm.get("bias")
        if w is None or g is None:
            w, g, b = self.weight, self.weight_scale, self.bias
        if (b is not None and not bias_deferred) or dtype != torch.float16 or g.dtype != torch.float16:
            return False
        if not w.is_cuda or w.stride(-1) != 1 or w.stride(0) % 16 or w.data_ptr() % 16:
            return False                                   # what the C side requires of the packed rows for the flag
        # (asked once per projection and forward by the fused prompt pass: 221 ctypes calls + device guards per 7B forward, 2 ms of
        #  its 6.3 ms host time, before the answer was cached per row count and device)
        cache = self.__dict__.get("_pres_ok")
        if cache is None:
            cache = self.__dict__["_pres_ok"] = {}
        key = (T, w.device.index)
        ok = cache.get(key)
        if ok is None:
            if len(cache) > 256:
                cache.clear()
            with torch.cuda.device(w.device):              # eligibility depends on the CU count of the device that will run it
                ok = cache[key] = bool(_lib.load().onebit_linear_prescaled_ok(T, self.in_features, self.out_features, _dtype_code(dtype)))
        return ok

    def pre_layernorm_prescaled(self, a: torch.Tensor) -> torch.Tensor:
        """``pre_layernorm`` on activations the producer already scaled: ``a = fp16(x * input_factor)``
        (onebit_rows_res_ln_rms / onebit_rows_swiglu with ``h_next``), ONEBIT_FLAG_PRESCALED.  Only where
        ``prescaled_ok(T)``; the C side refuses the flag elsewhere."""
        if self.bias is not None:
            raise RuntimeError("pre_layernorm: the bias is added after the LayerNorm; use forward")
        return self.forward(a, _pre_ln=True, _prescaled=True)

    def forward(self, input: torch.Tensor, _pre_ln: bool = False, _prescaled: bool = False) -> torch.Tensor:
        # (Host cost matters here: the reference's model code calls this 224 times per decoded token and the kernels behind a
        #  single-token call take ~10 us.  Parameters are read from _parameters directly -- nn.Module.__getattr__ costs 0.2 us per
        #  access --, the device guard is taken only when the input lives on another device than the current one, the stream comes
        #  from the raw-stream accessor: 25.4 -> 14.9 us per call at T = 1, tools/module_overhead_probe.py.)
        K, N = self.in_features, self.out_features
        if input.shape[-1] != K:
            raise RuntimeError(f"BitLinearInf: expected last dim {K}, got {tuple(input.shape)}")
        prm = self._parameters
        w, h, g, b = prm.get("weight"), prm.get("input_factor"), prm.get("weight_scale"), prm.get("bias")
        if w is None or h is None or g is None:                # (someone re-registered them as buffers / plain attributes)
            w, h, g, b = self.weight, self.input_factor, self.weight_scale, self.bias
        _require_gpu(input, "BitLinearInf.forward")
        _require_gpu(w, "BitLinearInf.forward (parameters)")
        pdt = g.dtype
        # bitnet.py:113 multiplies input by input_factor (type promotion), :115 then needs the
        # product to have the dtype of the unpacked weight (= weight_scale.dtype).
        cdt = input.dtype if input.dtype == h.dtype else torch.promote_types(input.dtype, h.dtype)
        if cdt != pdt:
            raise RuntimeError(
                f"BitLinearInf: input dtype {input.dtype} with parameters of dtype {pdt} "
                f"(the reference's F.linear raises on this mix as well)")
        code = _dtype_code(cdt)
        ln = self._modules.get("layernorm")
        if _pre_ln or isinstance(ln, nn.Identity):
            flags, eps = _lib.FLAG_SKIP_LN, 0.0
        elif isinstance(ln, nn.LayerNorm) and not ln.elementwise_affine:
            flags, eps = 0, float(ln.eps)
        else:
            raise RuntimeError("BitLinearInf.layernorm must be the parameter-free LayerNorm or nn.Identity")

        x = input if input.dtype == cdt else input.to(cdt)
        x = x.reshape(-1, K)
        if not x.is_contiguous():
            x = x.contiguous()
        T = x.shape[0]
        if w.dim() != 2 or w.shape[0] != N or w.shape[1] != K // 8 or w.dtype not in (torch.int8, torch.uint8):
            raise RuntimeError(f"BitLinearInf.weight must be int8 [{N}, {K // 8}], got {w.dtype} {tuple(w.shape)}")
        if w.stride(1) != 1:
            w = w.contiguous()
        if not h.is_contiguous():
            h = h.contiguous()
        if not g.is_contiguous():
            g = g.contiguous()
        if b is not None:
            b = b.to(cdt).contiguous()
        dev = x.device
        y = torch.empty((T, N), dtype=cdt, device=dev)
        lib = _lib.load()
        guard = None
        if dev.index is not None and dev.index != torch.cuda.current_device():
            guard = torch.cuda.device(dev)
            guard.__enter__()
        try:
            if _prescaled:
                flags |= _lib.FLAG_PRESCALED
                ws_bytes = 0                                   # the scaled rows ARE the input
            else:
                cache = self.__dict__.get("_ws_bytes")
                if cache is None:
                    cache = self.__dict__["_ws_bytes"] = {}
                key = (T, code, dev.index)
                ws_bytes = cache.get(key)
                if ws_bytes is None:                           # (depends on the shape and the device's CU count only)
                    if len(cache) > 256:
                        cache.clear()
                    ws_bytes = cache[key] = int(lib.onebit_linear_workspace_bytes(T, K, N, code))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
            rc = lib.onebit_linear_forward(
                w.data_ptr(), w.stride(0), x.data_ptr(), h.data_ptr(), g.data_ptr(),
                None if b is None else b.data_ptr(), y.data_ptr(), None,
                None if ws is None else ws.data_ptr(), ws_bytes,
                T, K, N, code, eps, flags, _stream_ptr(dev))
        finally:
            if guard is not None:
                guard.__exit__(None, None, None)
        if rc:
            _lib.check(rc, "onebit_linear_forward")
        return y.view(*input.shape[:-1], N)


# the name BASELINE.json uses for the layer
OneBitLinear = BitLinearInf
