#!/usr/bin/env python3
"""pack_nonsign.npz: the reference's ``fp16_to_int8`` (scripts/convert_llama_to_infer_ckpt.py:7-15) on tensors
that are NOT pure +-1 -- its asserts are commented out (:8-9), so it accepts them.  What it computes there:
``v = (0 - s + 1) / 2`` in the tensor's dtype, truncated to uint8 (``.to(torch.uint8)``), and the byte is the
uint8 matmul ``sum_i v_i * 2^i`` (wraps mod 256) -- so |s| > 1 spills into the neighbouring bit positions.
The fixture pins ``onebit_fp16_to_int8`` / ``ob_oracle_fp16_to_int8`` on that domain (s <= 1, i.e. v >= 0:
negative v -> uint8 is implementation-defined in torch and is left out).

Runs only in the build container (imports the reference read-only; only the function's AST node is executed,
no reference text is written).  Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens_pack_nonsign.py
"""
import os

import numpy as np
import torch

from gen_goldens import load_ref_packer

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    f = load_ref_packer()
    g = torch.Generator().manual_seed(99)
    vals = torch.tensor([-509.0, -255.0, -7.0, -3.0, -2.5, -2.0, -1.5, -1.0, -0.75, -0.5, -0.25, -0.0, 0.0, 0.25, 0.5, 0.999, 1.0])
    out = {}
    for i, (N, K) in enumerate([(4, 8), (6, 64), (16, 256)]):
        s = vals[torch.randint(0, len(vals), (N, K), generator=g)]
        s[0, :8] = torch.tensor([-3.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, -3.0])        # spill out of bit 7 wraps away
        for dt, name in ((torch.float32, "f32"), (torch.float16, "f16")):
            p = f(s.to(dt))
            assert p.dtype == torch.int8 and p.shape == (N, K // 8)
            out[f"s_{i}"] = s.numpy().astype(np.float32)
            out[f"packed_{name}_{i}"] = p.numpy()
    out["n_cases"] = np.array(3)
    np.savez_compressed(os.path.join(OUT, "pack_nonsign.npz"), **out)
    print("wrote pack_nonsign.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
