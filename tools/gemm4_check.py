"""gemm4 (32x32x16 MFMA) against gemm3 (16x16x32): bit-identity of the pre-LayerNorm outputs on several shapes (same k order inside a
step?  no -- the two kernels add the same products in different orders: compare within fp32 round-off) and the time per call.
Run twice: OB_GEMM4=0 / 1 write /tmp/g_<v>.pt; this script compares when both exist."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf, _lib
from onebit_amd.bitnet import _stream_ptr
dev = torch.device("cuda:0")
lib = _lib.load()
v = os.environ.get("OB_GEMM4", "0")
outs = {}
g = torch.Generator(device="cpu").manual_seed(1)
for (T, K, N) in [(16384, 4096, 11008), (16384, 11008, 4096), (16384, 4096, 4096), (4120, 5120, 13824), (4120, 13824, 5120), (2048, 4096, 4096), (700, 4096, 11004), (513, 1024, 2000)]:
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    x = torch.randn(T, K, generator=g).half().to(dev)
    y = torch.empty(T, N, device=dev, dtype=torch.float16)
    wsb = lib.onebit_linear_workspace_bytes(T, K, N, 0)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.onebit_linear_forward(m.weight.data_ptr(), m.weight.stride(0), x.data_ptr(), m.input_factor.data_ptr(), m.weight_scale.data_ptr(), None,
                                             y.data_ptr(), None, ws.data_ptr() if wsb else None, wsb, T, K, N, 0, 1e-5, 1, _stream_ptr(dev)), "fwd")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("OB_GEMM4=%s T=%5d K=%5d N=%5d  %.3f ms = %.1f TFLOP/s (incl. the scale pass)" % (v, T, K, N, ms, 2.0 * T * K * N / ms / 1e9), flush=True)
    outs[(T, K, N)] = y.cpu()
torch.save(outs, "/tmp/g_%s.pt" % v)
if os.path.exists("/tmp/g_0.pt") and os.path.exists("/tmp/g_1.pt"):
    a, b = torch.load("/tmp/g_0.pt"), torch.load("/tmp/g_1.pt")
    for k in a:
        d = (a[k].float() - b[k].float()).abs()
        ulp = d / (a[k].float().abs().clamp_min(1e-3) * 2.0 ** -10)
        print(k, "max |diff| %.4g  max ulps %.2f  differing %.4f %%" % (float(d.max()), float(ulp.max()), 100.0 * float((a[k] != b[k]).float().mean())))
