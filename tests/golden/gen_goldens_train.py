#!/usr/bin/env python3
"""Golden vectors of the reference's train-mode BitLinear (bitnet.py:14-68, imported by path; build
container only): forward output and the gradients of a scalar loss w.r.t. input, latent weight,
weight_scale, input_factor and bias, on a case whose latent weight contains exact zeros.
Output: tests/golden/train_bitlinear.npz.   Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens_train.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/transformers/src/transformers/models/bitnet.py"
if not os.path.exists(REF):
    sys.exit("reference not present")
spec = importlib.util.spec_from_file_location("ref_bitnet", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

g = torch.Generator().manual_seed(17)
K, N, T = 64, 24, 5
m = ref.BitLinear(K, N, bias=True)
with torch.no_grad():
    w = 0.3 * torch.randn(N, K, generator=g)
    w[0, :4] = 0.0
    w[5, 7] = 0.0
    m.weight.copy_(w)
    m.weight_scale.copy_(0.1 * (0.5 + torch.rand(N, generator=g)) * torch.where(torch.rand(N, generator=g) < 0.2, -1.0, 1.0))
    m.input_factor.copy_(0.1 * (0.5 + torch.rand(K, generator=g)))
    m.bias.copy_(0.1 * torch.randn(N, generator=g))
x = torch.randn(T, K, generator=g, requires_grad=True)
coef = torch.randn(T, N, generator=g)
y = m(x)
(y * coef).sum().backward()
out = {"K": np.array(K), "N": np.array(N), "x": x.detach().numpy(), "coef": coef.numpy(), "y": y.detach().numpy(),
       "gx": x.grad.numpy()}
for n, p in m.named_parameters():
    out["p_" + n], out["g_" + n] = p.detach().numpy(), p.grad.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_bitlinear.npz"), **out)
print("wrote train_bitlinear.npz", {k: v.shape for k, v in out.items()})
