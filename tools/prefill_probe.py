"""Prefill layer benchmark (BASELINE config 3): [8, 2048, 4096] -> 11008, HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf, _lib
from onebit_amd.bitnet import _stream_ptr
if os.environ.get("OB_LIB"):
    _lib.LIB_PATH = os.environ["OB_LIB"]
elif os.environ.get("OB_EXTRA"):
    import subprocess
    so = "/tmp/libonebit_exp.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
                           *os.environ["OB_EXTRA"].split(), "-o", so, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "onebit_amd/csrc/onebit_hip.hip"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "onebit_amd/csrc/onebit_mixed.hip")])
    _lib.LIB_PATH = so
dev = torch.device("cuda:0")
lib = _lib.load()
SHAPES = [(16384, 4096, 11008)] if (os.environ.get("OB_EXTRA") or os.environ.get("OB_ONE")) else None
for (T, K, N) in SHAPES or [(16384, 4096, 11008), (16384, 11008, 4096), (16384, 4096, 4096), (2048, 4096, 11008), (256, 4096, 11008), (64, 4096, 11008), (32, 4096, 11008), (16, 4096, 11008), (8, 4096, 11008), (2, 4096, 11008), (32, 11008, 4096), (32, 4096, 4096), (16, 4096, 4096)]:
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), dtype=torch.uint8, device=dev).view(torch.int8)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, device=dev))).half()
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, device=dev))).half()
    x = torch.randn(T, K, device=dev).half()
    y = torch.empty(T, N, device=dev, dtype=torch.float16)
    wsb = 0 if os.environ.get("OB_NO_WS") else lib.onebit_linear_workspace_bytes(T, K, N, 0)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    def run(flags):
        rc = lib.onebit_linear_forward(m.weight.data_ptr(), m.weight.stride(0), x.data_ptr(), m.input_factor.data_ptr(),
                                       m.weight_scale.data_ptr(), None, y.data_ptr(), None, ws.data_ptr() if wsb else None, wsb, T, K, N, 0, 1e-5, flags, _stream_ptr(dev))
        _lib.check(rc, "fwd")
    res = []
    for flags in (1, 0):          # 1 = skip LayerNorm (GEMM only), 0 = full forward
        for _ in range(3): run(flags)
        torch.cuda.synchronize()
        n = 10
        graph = None
        if T <= 256:              # small launches: time graph replays (the Python launch path costs ~10 us per call)
            n = 50
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(n): run(flags)
            graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(n): run(flags)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    fl = 2.0 * T * K * N
    print("T=%5d K=%5d N=%5d  gemm %.3f ms = %.1f TFLOP/s   gemm+LN %.3f ms = %.1f TFLOP/s" % (T, K, N, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9))
