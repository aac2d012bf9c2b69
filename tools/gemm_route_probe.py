"""Which prefill GEMM route is faster at medium T: default dispatch vs the LDS-DMA kernel forced (OB_GEMM3=2)?
Times onebit_linear_forward (SKIP_LN, with workspace: the pre-scaling pass is included) per shape.  Run twice:
  python tools/gemm_route_probe.py ; OB_GEMM3=2 python tools/gemm_route_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for T, K, N in ((512, 4096, 11008), (1024, 4096, 11008), (1024, 4096, 4096), (2048, 4096, 4096), (2048, 11008, 4096), (4096, 4096, 4096), (1536, 4096, 11008), (3072, 4096, 4096)):
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(T, K, generator=g).half().to(dev)
    for _ in range(3): u = m.pre_layernorm(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): u = m.pre_layernorm(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("OB_GEMM3=%s T=%5d %5d->%5d: %.3f ms = %.0f TFLOP/s  (prescaled_ok %s)" % (os.environ.get("OB_GEMM3", "-"), T, K, N, ms, 2.0 * T * K * N / ms / 1e9, m.prescaled_ok(T)), flush=True)
    del m, x, u
