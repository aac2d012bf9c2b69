#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container, where /root/reference exists (it does not
exist on the GPU box and nothing at test/bench time reads it).  The reference
is imported read-only:

* ``BitLinearInf`` from transformers/src/transformers/models/bitnet.py (loaded
  by file path; it depends on torch only);
* ``fp16_to_int8`` from scripts/convert_llama_to_infer_ckpt.py.  That script
  executes a hard-coded conversion at import time, so only the function's own
  AST node is compiled and executed here, in memory; no reference source text
  is written anywhere.

Outputs are data only (inputs + the reference's outputs), a few hundred KB:

  pack.npz      sign tensors and the bytes fp16_to_int8 returns for them
  unpack.npz    int8_to_fp16 over all 256 byte values
  forward.npz   BitLinearInf.forward (fp32 and fp16 parameters) on small
                shapes / edge cases: post-LN y and pre-LN u (layernorm swapped
                for nn.Identity on the imported module)

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens.py
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref_bitnet():
    spec = importlib.util.spec_from_file_location(
        "ref_bitnet", f"{REF}/transformers/src/transformers/models/bitnet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_packer():
    path = f"{REF}/scripts/convert_llama_to_infer_ckpt.py"
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "fp16_to_int8"]
    assert len(fn) == 1
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns["fp16_to_int8"]


def gen_pack(fp16_to_int8):
    g = torch.Generator().manual_seed(1234)
    out = {}
    cases = [(3, 8), (5, 32), (16, 64), (48, 256), (7, 96)]
    for i, (N, K) in enumerate(cases):
        s = torch.where(torch.rand(N, K, generator=g) < 0.5, -1.0, 1.0)
        # engineered bytes in row 0: 0x00, 0x01, 0x80, 0xFF (LSB-first)
        if K >= 32:
            s[0, 0:8] = 1.0
            s[0, 8:16] = 1.0; s[0, 8] = -1.0
            s[0, 16:24] = 1.0; s[0, 23] = -1.0
            s[0, 24:32] = -1.0
        # exact zeros (what torch.sign gives for a latent weight of 0) -> +1
        s[N - 1, 1] = 0.0
        s[N - 1, K - 1] = 0.0
        for dt, name in ((torch.float32, "f32"), (torch.float16, "f16")):
            packed = fp16_to_int8(s.to(dt))
            assert packed.dtype == torch.int8 and packed.shape == (N, K // 8)
            out[f"signs_{i}"] = s.numpy().astype(np.float32)
            out[f"packed_{name}_{i}"] = packed.numpy()
    # latent weights through torch.sign first, as the converter main loop does
    w = torch.randn(9, 64, generator=g)
    w[2, 5] = 0.0; w[2, 6] = -0.0; w[3, 0] = 1e-30; w[3, 1] = -1e-30
    out["latent_w"] = w.numpy()
    out["latent_packed"] = fp16_to_int8(torch.sign(w)).numpy()
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(f"{OUT}/pack.npz", **out)
    print("pack.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gen_unpack(ref):
    out = {}
    for dt, name in ((torch.float32, "f32"), (torch.float16, "f16")):
        lin = ref.BitLinearInf(2048, 1, dtype=dt)
        allbytes = torch.arange(256, dtype=torch.uint8).view(torch.int8).view(1, 256)
        out[f"dense_{name}"] = lin.int8_to_fp16(allbytes).float().numpy()
    out["bytes"] = np.arange(256, dtype=np.uint8).view(np.int8).reshape(1, 256)
    np.savez_compressed(f"{OUT}/unpack.npz", **out)
    print("unpack.npz ok")


def make_layer(ref, K, N, dt, g, bias, special):
    lin = ref.BitLinearInf(K, N, bias=bias, dtype=dt)
    lin.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8)
    h = 0.5 + torch.rand(K, generator=g)
    gs = 0.5 + torch.rand(N, generator=g)
    if special == "mixed":            # mixed signs and an exact zero entry
        h = h * torch.where(torch.rand(K, generator=g) < 0.2, -1.0, 1.0)
        gs = gs * torch.where(torch.rand(N, generator=g) < 0.2, -1.0, 1.0)
        h[K // 3] = 0.0
        gs[N // 2] = 0.0
    if special == "small":            # trained-checkpoint-like magnitudes
        h = h * 0.1
        gs = gs * 0.1
    lin.input_factor.data = h.to(dt)
    lin.weight_scale.data = gs.to(dt)
    if bias:
        lin.bias.data = (0.1 * torch.randn(N, generator=g)).to(dt)
    return lin


def gen_forward(ref):
    g = torch.Generator().manual_seed(4321)
    out = {}
    idx = 0
    shapes = [(32, 8), (64, 48), (256, 688), (688, 256), (1024, 512)]
    leads = [(1,), (1, 1), (2, 5)]
    meta = []
    for (K, N) in shapes:
        for li, lead in enumerate(leads):
            for dt, dname in ((torch.float32, "f32"), (torch.float16, "f16")):
                special = ["plain", "mixed", "small"][(idx // 2) % 3]
                bias = (idx // 2) % 4 == 1
                lin = make_layer(ref, K, N, dt, g, bias, special)
                x = torch.randn(*lead, K, generator=g).to(dt)
                with torch.no_grad():
                    y = lin(x)
                    ln = lin.layernorm
                    lin.layernorm = torch.nn.Identity()
                    b = lin.bias
                    lin.bias = None
                    u = lin(x)
                    lin.layernorm, lin.bias = ln, b
                assert y.dtype == dt and y.shape == (*lead, N)
                p = f"c{idx}_"
                out[p + "packed"] = lin.weight.data.numpy()
                out[p + "x"] = x.numpy(); out[p + "h"] = lin.input_factor.data.numpy()
                out[p + "g"] = lin.weight_scale.data.numpy()
                if bias:
                    out[p + "bias"] = lin.bias.data.numpy()
                out[p + "y"] = y.numpy(); out[p + "u"] = u.numpy()
                meta.append((idx, K, N, dname, special, int(bias), len(lead)))
                idx += 1
    # constant-row case: all z equal -> variance ~ 0, pins the eps handling.
    # weight bytes 0 (= all +1, the reset_parameters state, bitnet.py:89-92).
    for dt, dname in ((torch.float32, "f32"), (torch.float16, "f16")):
        lin = ref.BitLinearInf(64, 16, dtype=dt)           # g = h = 1, W = +1
        x = torch.randn(2, 64, generator=g).to(dt)
        with torch.no_grad():
            y = lin(x)
            lin.layernorm = torch.nn.Identity()
            u = lin(x)
        p = f"c{idx}_"
        out[p + "packed"] = lin.weight.data.numpy()
        out[p + "x"] = x.numpy(); out[p + "h"] = lin.input_factor.data.numpy()
        out[p + "g"] = lin.weight_scale.data.numpy()
        out[p + "y"] = y.numpy(); out[p + "u"] = u.numpy()
        meta.append((idx, 64, 16, dname, "reset", 0, 1))
        idx += 1
    out["meta_idx_K_N_bias_nlead"] = np.array([(m[0], m[1], m[2], m[5], m[6]) for m in meta])
    out["meta_dtype"] = np.array([m[3] for m in meta])
    out["meta_special"] = np.array([m[4] for m in meta])
    np.savez_compressed(f"{OUT}/forward.npz", **out)
    print("forward.npz cases:", idx, "bytes:", os.path.getsize(f"{OUT}/forward.npz"))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures can only be regenerated in the build container")
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref = load_ref_bitnet()
    gen_pack(load_ref_packer())
    gen_unpack(ref)
    gen_forward(ref)
