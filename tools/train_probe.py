"""Times the train-mode BitLinear layer (forward + backward, fp16) at a 7B MLP shape: 3 GEMMs of 2 T K N flops each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd.train import BitLinear
dev = "cuda:0"
T, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 11008)
torch.manual_seed(0)
m = BitLinear(K, N, dtype=torch.float16).to(dev)
with torch.no_grad():
    m.weight.normal_(0, 0.02); m.weight_scale.uniform_(0.05, 0.15); m.input_factor.uniform_(0.05, 0.15)
x = torch.randn(T, K, device=dev, dtype=torch.float16, requires_grad=True)
gy = torch.randn(T, N, device=dev, dtype=torch.float16)
def step():
    y = m(x)
    y.backward(gy)
    m.zero_grad(set_to_none=True); x.grad = None
for _ in range(2): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("train layer T=%d K=%d N=%d fp16: forward + backward %.2f ms = %.0f TFLOP/s (3 GEMMs)" % (T, K, N, ms, 6.0 * T * K * N / ms / 1e9))
