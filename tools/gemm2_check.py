"""A/B of the two prefill GEMM kernels on the same inputs: determinism of each, element-wise agreement."""
import os, sys, subprocess
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else None
if mode is None:
    outs = {}
    for g in ("0", "2"):
        env = dict(os.environ, OB_GEMM2=g)
        subprocess.check_call([sys.executable, __file__, g], env=env)
        outs[g] = np.load("/tmp/gemm2_%s.npy" % g)
    a, b = outs["0"].astype(np.float32), outs["2"].astype(np.float32)
    d = np.abs(a - b); ulp = np.maximum(np.abs(a), 2.0 ** -14) * 2.0 ** -10
    print("differ: %.4f%%  max |d|/ulp %.2f  >2ulp: %d of %d" % (100 * (a != b).mean(), (d / ulp).max(), int((d > 2.001 * ulp).sum()), a.size))
    bad = np.argwhere(d > 2.001 * ulp)[:10]
    for t, n in bad: print("  t=%d n=%d old=%r new=%r" % (t, n, a[t, n], b[t, n]))
else:
    from onebit_amd import BitLinearInf
    dev = torch.device("cuda:0")
    K, N, T = 4096, 11008, 16384
    g = torch.Generator(device="cpu").manual_seed(21)
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(T, K, generator=g).half().to(dev)
    m.layernorm = torch.nn.Identity()
    u1 = m(x).clone(); u2 = m(x).clone(); u3 = m(x)
    print("OB_GEMM2=%s deterministic: %s %s" % (mode, torch.equal(u1, u2), torch.equal(u1, u3)))
    np.save("/tmp/gemm2_%s.npy" % mode, u1[::7].cpu().numpy())
