"""Times onebit_attention_prefill against torch SDPA (AOTriton) at the BASELINE config-3 attention shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.llama import hip_attention_prefill
dev = "cuda:0"
B, S, H, D = 8, 2048, 32, 128
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn(B, S, H, D, generator=g, device=dev, dtype=torch.float16)
k = torch.randn(B, H, S, D, generator=g, device=dev, dtype=torch.float16)
v = torch.randn(B, H, S, D, generator=g, device=dev, dtype=torch.float16)
flop = 4.0 * B * H * S * S * D / 2
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_hip = timeit(lambda: hip_attention_prefill(q, k, v, 0))
qt = q.transpose(1, 2)
t_sdpa = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qt, k, v, is_causal=True))
print("hip  %.3f ms  %.0f TFLOP/s (causal flops)" % (t_hip, flop / t_hip / 1e9))
print("sdpa %.3f ms  %.0f TFLOP/s" % (t_sdpa, flop / t_sdpa / 1e9))
