"""Child process of test_large_tile_kernels_forced_on_ragged_shapes: the dispatch thresholds of the two
256 x 256 prefill kernels are read once per process from the environment (OB_GEMM2 / OB_GEMM3 = 2 forces
the kernel on any eligible shape), so the forced runs live in their own interpreter.  Compares the HIP
path with the oracle on shapes whose last tiles are ragged in T and N; exits non-zero on a mismatch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.oracle import COracle          # noqa: E402  (test infrastructure: the checker)
from test_gpu_parity import _check_f16, _make_layer, _t   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    co = COracle()
    rng = np.random.default_rng(77)
    # (T, K, N): one full + one ragged tile each way; a single ragged tile; K = 1, 2 and 9 quads of steps
    for (T, K, N) in ((300, 512, 260), (192, 256, 40), (257, 2304, 516), (513, 1024, 256)):
        packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
        h = (0.1 * (0.5 + rng.random(K)) * np.where(rng.random(K) < 0.1, -1, 1)).astype(np.float16)
        g = (0.1 * (0.5 + rng.random(N)) * np.where(rng.random(N) < 0.1, -1, 1)).astype(np.float16)
        x = rng.standard_normal((T, K)).astype(np.float16)
        y_ref, u_ref = co.forward_f16(packed, x, h, g, None, return_pre_ln=True)
        m = _make_layer(K, N, torch.float16, dev, packed, h, g)
        y = m(_t(x, dev)).cpu().numpy()
        m.layernorm = torch.nn.Identity()
        u = m(_t(x, dev)).cpu().numpy()
        _check_f16(y, u, y_ref, u_ref, (T, K, N))
    print("forced-route ok")


if __name__ == "__main__":
    main()
