"""Race screen for the LDS-DMA prefill GEMM: the same call repeated many times on several shapes (whole-chip
grids and forced small ones) must give bit-identical outputs every time -- a DMA read placed by luck instead
of by the wait / barrier count shows up as rare differing tiles.  Usage: [OB_GEMM3=2] python tools/gemm3_race_screen.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
shapes = [(16384, 4096, 11008), (16384, 11008, 4096), (2048, 4096, 11008), (8192, 5120, 13824)]
if os.environ.get("OB_GEMM3") == "2":
    shapes += [(300, 512, 260), (257, 2304, 516), (513, 1024, 256), (192, 256, 40)]
bad = 0
for (T, K, N) in shapes:
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    m.layernorm = torch.nn.Identity()
    x = torch.randn(T, K, generator=g).half().to(dev)
    ref = m(x).clone()
    n = 60 if T * N < 5e7 else 25
    diff = 0
    for i in range(n):
        # a competing stream of memory traffic between repetitions changes the DMA arrival pattern
        if i % 3 == 0:
            junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev).random_()
        y = m(x)
        if not torch.equal(y, ref):
            diff += 1
    print("T=%5d K=%5d N=%5d: %d / %d repetitions differ" % (T, K, N, diff, n))
    bad += diff
    del m, x, ref
print("race screen:", "CLEAN" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
