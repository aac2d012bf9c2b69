"""Train-mode BitLinear / SignSTE torch restatement (oracle/train_ref.py, the checker of the HIP layer) against forward outputs and gradients recorded
from the reference's own class (tests/golden/gen_goldens_train.py), and its relation to the packed layer."""
import os

import numpy as np
import torch

from oracle.train_ref import BitLinear


def _load(golden_dir):
    z = np.load(os.path.join(golden_dir, "train_bitlinear.npz"))
    m = BitLinear(int(z["K"]), int(z["N"]), bias=True)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(z["p_" + n]))
    return z, m


def test_forward_and_ste_gradients_match_reference(golden_dir):
    z, m = _load(golden_dir)
    assert [n for n, _ in m.named_parameters()] == ["weight", "weight_scale", "input_factor", "bias"]
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y = m(x)
    assert np.abs(y.detach().numpy() - z["y"]).max() <= 1e-5
    (y * torch.from_numpy(z["coef"])).sum().backward()
    assert np.abs(x.grad.numpy() - z["gx"]).max() <= 1e-5 * max(1.0, np.abs(z["gx"]).max())
    for n, p in m.named_parameters():
        ref = z["g_" + n]
        assert np.abs(p.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), n


def test_packed_layer_differs_only_at_zero_latent_weights(golden_dir, coracle):
    """sign(0) = 0 in training, +1 after packing (convert_llama_to_infer_ckpt.py:10): the oracle's
    inference forward on the packed signs equals the train-mode forward once the zeros of the
    latent weight are replaced by a positive value, and differs before."""
    from oracle.oracle import np_pack_signs
    z, m = _load(golden_dir)
    x = z["x"]
    w = z["p_weight"].copy()
    packed = np_pack_signs(np.sign(w))
    y_inf = coracle.forward_f32(packed, x, z["p_input_factor"], z["p_weight_scale"], z["p_bias"])
    with torch.no_grad():
        y_train = m(torch.from_numpy(x)).numpy()
        assert np.abs(y_inf[:, :] - y_train).max() > 1e-3              # rows 0 and 5 hold zero weights
        m.weight[m.weight == 0] = 1e-3
        y_fixed = m(torch.from_numpy(x)).numpy()
    assert np.abs(y_inf - y_fixed).max() <= 2e-4
