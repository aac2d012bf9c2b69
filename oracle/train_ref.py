"""Train-mode 1-bit linear layer, torch-ops restatement -- TEST INFRASTRUCTURE (part of the oracle).

Host-side statement of the reference's ``BitLinear`` / ``SignSTE``
(``transformers/src/transformers/models/bitnet.py:14-68``) with plain torch ops and torch autograd: latent
full-precision weight, ``sign`` in the forward pass with the straight-through estimator
``grad * (1.001 - tanh(w)^2)`` in the backward pass (``:21-23``), the same
``(x * input_factor) @ sign(W)^T * weight_scale -> LayerNorm (+ bias)`` pipeline as the packed layer.
Pinned by ``tests/golden/train_bitlinear.npz`` (forward and gradients recorded from the reference class,
``tests/golden/gen_goldens_train.py``).  It is the checker of the HIP train-mode layer
(``onebit_amd/train.py`` -> ``onebit_train_forward`` / ``onebit_train_backward``) at sizes the fixture does not
cover; only tests import it, the product never does.
"""
from __future__ import annotations

import math

import torch
from torch import nn
import torch.nn.functional as F


class _SignSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w):
        ctx.save_for_backward(w)
        return torch.sign(w)

    @staticmethod
    def backward(ctx, grad):
        (w,) = ctx.saved_tensors
        return grad * (1.001 - torch.tanh(w) ** 2)


class SignSTE(nn.Module):
    def forward(self, w):
        return _SignSTE.apply(w)


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
        x = x * self.input_factor.view(1, self.in_features)
        out = F.linear(x, self.sign(self.weight))
        out = out * self.weight_scale.view(1, self.out_features)
        out = self.layernorm(out)
        if self.bias is not None:
            out = out + self.bias
        return out
