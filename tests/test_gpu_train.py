"""Train-mode BitLinear on the GPU (onebit_amd/train.py -> onebit_train_forward / onebit_train_backward, HIP MFMA
kernels) against forward outputs and gradients recorded from the REFERENCE class (bitnet.py:14-68;
tests/golden/train_bitlinear.npz) and, at sizes the fixture does not cover, against the torch-ops restatement
pinned by the same fixture (oracle/train_ref.py, evaluated in float64 on the CPU)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_grads(K, N, params, x, coef, bias=True, dtype=torch.float64):
    from oracle.train_ref import BitLinear as RefBitLinear
    m = RefBitLinear(K, N, bias=bias, dtype=dtype)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(params[n].to(dtype))
    xr = x.detach().to(dtype).cpu().requires_grad_(True)
    y = m(xr)
    (y * coef.to(dtype).cpu()).sum().backward()
    return y.detach(), xr.grad, {n: p.grad for n, p in m.named_parameters()}


def test_forward_and_gradients_match_reference_fixture(golden_dir):
    from onebit_amd.train import BitLinear
    z = np.load(os.path.join(golden_dir, "train_bitlinear.npz"))
    K, N = int(z["K"]), int(z["N"])
    m = BitLinear(K, N, bias=True).to(DEV)
    assert [n for n, _ in m.named_parameters()] == ["weight", "weight_scale", "input_factor", "bias"]
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(z["p_" + n]))
    assert (m.weight == 0).sum() == 5                       # the fixture's exact zeros: sign(0) = 0 on this path
    x = torch.from_numpy(z["x"]).to(DEV).requires_grad_(True)
    y = m(x)
    assert np.abs(y.detach().cpu().numpy() - z["y"]).max() <= 1e-5
    (y * torch.from_numpy(z["coef"]).to(DEV)).sum().backward()
    assert np.abs(x.grad.cpu().numpy() - z["gx"]).max() <= 1e-5 * max(1.0, np.abs(z["gx"]).max())
    for n, p in m.named_parameters():
        ref = z["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), n


@pytest.mark.parametrize("T,K,N,bias", [(77, 200, 136, True), (5, 24, 8, False), (300, 1024, 520, True), (64, 4096, 256, False)])
def test_fp32_ragged_shapes_vs_restatement(T, K, N, bias):
    from onebit_amd.train import BitLinear
    g = torch.Generator().manual_seed(T + K + N)
    params = {"weight": 0.3 * torch.randn(N, K, generator=g),
              "weight_scale": 0.1 * (0.5 + torch.rand(N, generator=g)) * torch.where(torch.rand(N, generator=g) < 0.2, -1.0, 1.0),
              "input_factor": 0.1 * (0.5 + torch.rand(K, generator=g))}
    params["weight"][0, :3] = 0.0
    if bias:
        params["bias"] = 0.1 * torch.randn(N, generator=g)
    x = torch.randn(2, T, K, generator=g)[0:1].reshape(1, T, K)          # 3-D input, leading dims kept
    coef = torch.randn(1, T, N, generator=g)
    m = BitLinear(K, N, bias=bias).to(DEV)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(params[n])
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    assert y.shape == (1, T, N)
    (y * coef.to(DEV)).sum().backward()
    y_ref, gx_ref, gp_ref = _ref_grads(K, N, params, x, coef, bias)
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
    assert rel(y.detach(), y_ref) <= 2e-5
    assert rel(xd.grad, gx_ref) <= 2e-4
    for n, p in m.named_parameters():
        assert rel(p.grad, gp_ref[n]) <= 2e-4, n
    # deterministic: a second backward pass gives the same bits
    xd2 = x.to(DEV).requires_grad_(True)
    m.zero_grad()
    (m(xd2) * coef.to(DEV)).sum().backward()
    assert torch.equal(xd2.grad, xd.grad)


@pytest.mark.parametrize("T,K,N", [(256, 1024, 768), (77, 200, 136), (5, 24, 8), (300, 1000, 520), (129, 130, 131)])
def test_fp16_against_restatement_in_fp64(T, K, N):
    """fp16 tensors (ob_tgemm128_f16_kernel: 128 x 128 tiles, MFMA 16x16x32 f16, fp32 accumulate): every tensor-level op rounds once
    to fp16, so the results sit within fp16 rounding of the exact (float64) evaluation on the same fp16 inputs.  The ragged shapes
    take the kernel's partial tiles, its element-wise operand fetch (rows that are not 16-byte aligned) and the zero fill behind
    the transpose reads."""
    from onebit_amd.train import BitLinear
    g = torch.Generator().manual_seed(4 + T)
    params = {"weight": (0.3 * torch.randn(N, K, generator=g)).half().float(),
              "weight_scale": (0.1 * (0.5 + torch.rand(N, generator=g))).half().float(),
              "input_factor": (0.1 * (0.5 + torch.rand(K, generator=g))).half().float(),
              "bias": (0.1 * torch.randn(N, generator=g)).half().float()}
    x = torch.randn(T, K, generator=g).half().float()
    coef = torch.randn(T, N, generator=g).half().float()
    m = BitLinear(K, N, bias=True, dtype=torch.float16).to(DEV)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(params[n].half())
    xd = x.half().to(DEV).requires_grad_(True)
    y = m(xd)
    assert y.dtype == torch.float16
    (y * coef.half().to(DEV)).sum().backward()
    y_ref, gx_ref, gp_ref = _ref_grads(K, N, params, x, coef, True)
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
    assert rel(y.detach(), y_ref) <= 2e-3
    assert rel(xd.grad, gx_ref) <= 4e-3
    for n, p in m.named_parameters():
        assert p.grad.dtype == torch.float16
        assert rel(p.grad, gp_ref[n]) <= 4e-3, n


def test_train_then_convert_matches_packed_layer():
    """The train-mode layer and the packed inference layer agree once the latent weights are packed
    (convert_llama_to_infer_ckpt.py:26-34): same y within fp32 round-off when no latent weight is exactly 0."""
    from onebit_amd import BitLinearInf, pack_signs
    from onebit_amd.train import BitLinear
    K, N, T = 512, 384, 9
    g = torch.Generator().manual_seed(8)
    m = BitLinear(K, N).to(DEV)
    with torch.no_grad():
        m.weight.copy_(0.2 * torch.randn(N, K, generator=g) + 1e-3)
        m.weight_scale.copy_(0.1 * (0.5 + torch.rand(N, generator=g)))
        m.input_factor.copy_(0.1 * (0.5 + torch.rand(K, generator=g)))
    assert (m.weight == 0).sum() == 0
    inf = BitLinearInf(K, N, dtype=torch.float32).to(DEV)
    inf.weight.data = pack_signs(m.weight.data)
    inf.weight_scale.data, inf.input_factor.data = m.weight_scale.data.clone(), m.input_factor.data.clone()
    x = torch.randn(T, K, generator=g).to(DEV)
    with torch.no_grad():
        assert float((m(x) - inf(x)).abs().max()) <= 2e-4


def test_cpu_tensors_raise_and_argument_errors():
    from onebit_amd import _lib
    from onebit_amd.train import BitLinear
    m = BitLinear(16, 8)
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 16))
    lib = _lib.load()
    assert lib.onebit_train_workspace_bytes(0, 8, 8, 0) == 0
    assert lib.onebit_train_forward(None, None, None, None, None, None, None, None, 4, 8, 8, 0, 1e-5, None) == -1
    assert lib.onebit_train_forward(None, None, None, None, None, None, None, None, 4, 8, 8, 7, 1e-5, None) == -4
