"""TFLOP/s of the LDS-DMA prefill GEMM (pre-scaling pass included, module path) on the three 7B shapes at T = 16384, for the
current OB_GEMM3_WT setting; bit-identity of the two workgroup forms is checked by comparing against a saved reference run
(pass a file name to save / compare).  Usage: [OB_GEMM3_WT=1|2] python tools/gemm3_cmp.py [ref.pt]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
T = 16384
outs = {}
for K, N in ((4096, 11008), (11008, 4096), (4096, 4096)):
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(T, K, generator=g).half().to(dev)
    a = x * m.input_factor.data
    for _ in range(3): u = m.pre_layernorm_prescaled(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): u = m.pre_layernorm_prescaled(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("WT=%s  %5d -> %5d  T=%d: %.3f ms = %.0f TFLOP/s (GEMM kernel alone, pre-scaled rows)" % (os.environ.get("OB_GEMM3_WT", "2"), K, N, T, ms, 2.0 * T * K * N / ms / 1e9), flush=True)
    outs["%d_%d" % (K, N)] = u[::997].cpu()
    del m, x, a, u
if len(sys.argv) > 1:
    if os.path.exists(sys.argv[1]):
        ref = torch.load(sys.argv[1])
        print("bit-identical to the saved run:", all(torch.equal(ref[k], outs[k]) for k in outs))
    else:
        torch.save(outs, sys.argv[1])
