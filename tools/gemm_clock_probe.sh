# sclk / power while the prefill GEMM loops (gemm3 vs gemm4): is the chip power-limited there?
for v in 0 1; do
  OB_GEMM4=$v python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from onebit_amd import BitLinearInf, _lib
from onebit_amd.bitnet import _stream_ptr
dev = torch.device("cuda:0"); lib = _lib.load()
T, K, N = 16384, 4096, 11008
m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
m.weight.data = torch.randint(0, 256, (N, K // 8), dtype=torch.uint8, device=dev).view(torch.int8)
m.input_factor.data = (0.1 * (0.5 + torch.rand(K, device=dev))).half(); m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, device=dev))).half()
x = torch.randn(T, K, device=dev).half(); y = torch.empty(T, N, device=dev, dtype=torch.float16)
a = (x * m.input_factor).contiguous()
def run():
    _lib.check(lib.onebit_linear_forward(m.weight.data_ptr(), m.weight.stride(0), a.data_ptr(), m.input_factor.data_ptr(), m.weight_scale.data_ptr(), None,
                                         y.data_ptr(), None, None, 0, T, K, N, 0, 1e-5, 1 | 4, _stream_ptr(dev)), "fwd")
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(50): run()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print("OB_GEMM4=%s: %.3f ms per call, %.1f TFLOP/s (GEMM alone, pre-scaled rows), %d calls" % (os.environ["OB_GEMM4"], dt / n * 1e3, 2.0 * T * K * N * n / dt / 1e12, n), flush=True)
PY
  sleep 2.5
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.4; done
  wait
done
