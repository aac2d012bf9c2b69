"""Train-mode 1-bit linear layer on MI355X -- SURVEY.md section 8 rows a8 / f4.

``BitLinear`` mirrors the reference class (``transformers/src/transformers/models/bitnet.py:31-68``): same
constructor, parameter names (``weight`` latent full precision ``[out, in]``, ``weight_scale``, ``input_factor``,
optional ``bias``) and state-dict keys, so a training checkpoint of the reference loads unchanged and
``checkpoint.convert_train_state_dict`` turns it into the packed inference layout.  Forward and backward run
hand-written HIP kernels through the C ABI (``onebit_train_forward`` / ``onebit_train_backward``,
``csrc/ob_train.h``): the sign is taken while the latent weights are staged for the MFMA GEMM (no dense +-1
matrix in HBM), the backward pass applies the reference's straight-through estimator
``grad * (1.001 - tanh(w)^2)`` (``SignSTEFunc.backward``, ``:21-23``) in the weight-gradient GEMM's epilogue.
There is no CPU fallback: CPU tensors raise.  (The torch-ops restatement used to check this lives in
``oracle/train_ref.py``, test infrastructure.)

It differs from the packed ``BitLinearInf`` exactly where the reference's two classes differ: a latent weight of
exactly 0 is ``sign(0) = 0`` here and ``+1`` after packing (``convert_llama_to_infer_ckpt.py:10``).
"""
from __future__ import annotations

import math

import torch
from torch.autograd.function import once_differentiable
from torch import nn

from . import _lib
from .bitnet import _dtype_code, _require_gpu, _stream_ptr

__all__ = ["BitLinear", "SignSTE", "bitlinear_train"]


class _BitLinearFn(torch.autograd.Function):
    """y = LayerNorm(g * ((x * h) @ sign(W)^T)) (+ bias) with the STE backward, one C-ABI call each way."""

    @staticmethod
    def forward(ctx, x, weight, weight_scale, input_factor, bias, eps):
        _require_gpu(x, "BitLinear.forward")
        _require_gpu(weight, "BitLinear.forward (parameters)")
        dt = weight.dtype
        if not (x.dtype == dt == weight_scale.dtype == input_factor.dtype) or (bias is not None and bias.dtype != dt):
            raise RuntimeError("BitLinear: input and parameters must share one dtype (float16 or float32)")
        code = _dtype_code(dt)
        N, K = weight.shape
        x2 = x.reshape(-1, K).contiguous()
        T = x2.shape[0]
        w, g, h = weight.contiguous(), weight_scale.contiguous(), input_factor.contiguous()
        b = None if bias is None else bias.contiguous()
        y = torch.empty((T, N), dtype=dt, device=x.device)
        z = torch.empty((T, N), dtype=dt, device=x.device)
        stats = torch.empty((T, 2), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            rc = lib.onebit_train_forward(x2.data_ptr(), w.data_ptr(), h.data_ptr(), g.data_ptr(), None if b is None else b.data_ptr(),
                                          y.data_ptr(), z.data_ptr(), stats.data_ptr(), T, K, N, code, eps, _stream_ptr(x.device))
        _lib.check(rc, "onebit_train_forward")
        ctx.save_for_backward(x2, w, g, h, z, stats)
        ctx.has_bias, ctx.code, ctx.lead = b is not None, code, x.shape[:-1]
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @once_differentiable               # the gradients come out of a C-ABI call: no graph behind them, no double backward
    def backward(ctx, gy):
        x2, w, g, h, z, stats = ctx.saved_tensors
        N, K = w.shape
        T = x2.shape[0]
        gy2 = gy.reshape(T, N).contiguous()
        dev, dt = gy2.device, w.dtype
        gx = torch.empty((T, K), dtype=dt, device=dev)
        gw = torch.empty((N, K), dtype=dt, device=dev)
        gh = torch.empty(K, dtype=dt, device=dev)
        gg = torch.empty(N, dtype=dt, device=dev)
        gb = torch.empty(N, dtype=dt, device=dev) if ctx.has_bias else None
        lib = _lib.load()
        nws = lib.onebit_train_workspace_bytes(T, K, N, ctx.code)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.onebit_train_backward(gy2.data_ptr(), x2.data_ptr(), w.data_ptr(), h.data_ptr(), g.data_ptr(), z.data_ptr(),
                                           stats.data_ptr(), gx.data_ptr(), gw.data_ptr(), gh.data_ptr(), gg.data_ptr(),
                                           None if gb is None else gb.data_ptr(), ws.data_ptr(), nws, T, K, N, ctx.code,
                                           _stream_ptr(dev))
        _lib.check(rc, "onebit_train_backward")
        # (one fused call computes all five; outputs autograd did not ask for are dropped, not returned as garbage)
        need = ctx.needs_input_grad
        return (gx.view(*ctx.lead, K) if need[0] else None, gw if need[1] else None, gg if need[2] else None,
                gh if need[3] else None, gb if (gb is not None and need[4]) else None, None)


def bitlinear_train(x, weight, weight_scale, input_factor, bias=None, eps: float = 1e-5):
    """Functional form of ``BitLinear.forward`` (differentiable w.r.t. x and all four parameters)."""
    return _BitLinearFn.apply(x, weight, weight_scale, input_factor, bias, eps)


class SignSTE(nn.Module):
    """Kept for state-dict / module-tree compatibility with the reference (``bitnet.py:26-28``): the sign and its
    straight-through gradient are applied inside the fused kernels, this module holds no state."""

    def forward(self, w):          # pragma: no cover - not on the fused path
        raise RuntimeError("SignSTE is fused into BitLinear's HIP kernels; call BitLinear.forward")


class BitLinear(nn.Module):
    """Constructor, parameter names and state-dict keys of the reference class (``bitnet.py:31-48``).
    As there, ``weight`` is left uninitialised: it is always loaded from a pretrained checkpoint."""

    def __init__(self, in_features, out_features, groups=1, bias=False, device=None, dtype=None):
        kw = {"device": device, "dtype": dtype}
        super().__init__()
        self.in_features, self.out_features, self.groups = in_features, out_features, groups
        self.weight = nn.Parameter(torch.empty((out_features, in_features), **kw))
        self.weight_scale = nn.Parameter(torch.empty(out_features, **kw))
        self.sign = SignSTE()
        self.input_factor = nn.Parameter(torch.empty(in_features, **kw))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, **kw))
        else:
            self.register_parameter("bias", None)
        self.layernorm = nn.LayerNorm(out_features, elementwise_affine=False)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.constant_(self.weight_scale, 1.0)
        nn.init.constant_(self.input_factor, 1.0)
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features) if self.in_features > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        if x.shape[-1] != self.in_features:
            raise RuntimeError(f"BitLinear: expected last dim {self.in_features}, got {tuple(x.shape)}")
        if not (isinstance(self.layernorm, nn.LayerNorm) and not self.layernorm.elementwise_affine):
            raise RuntimeError("BitLinear.layernorm must be the parameter-free LayerNorm")
        return _BitLinearFn.apply(x, self.weight, self.weight_scale, self.input_factor, self.bias, float(self.layernorm.eps))
