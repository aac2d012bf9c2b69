"""Context for the prefill numbers: the vendor library's dense fp16 GEMM (torch.matmul -> hipBLASLt /
rocBLAS) on the same shapes, i.e. what the reference's forward would cost AFTER its unpack."""
import torch
dev = torch.device("cuda:0")
for (T, K, N) in [(16384, 4096, 11008), (16384, 11008, 4096), (16384, 4096, 4096), (2048, 4096, 11008)]:
    x = torch.randn(T, K, device=dev).half()
    w = torch.randn(N, K, device=dev).half()
    for _ in range(5): torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): torch.nn.functional.linear(x, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("dense fp16 F.linear T=%5d K=%5d N=%5d: %.3f ms = %.1f TFLOP/s" % (T, K, N, ms, 2.0 * T * K * N / ms / 1e9))
