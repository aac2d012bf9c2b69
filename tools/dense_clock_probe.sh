# The vendor library's dense fp16 GEMM (torch.matmul -> hipBLASLt) on the prefill shape, SUSTAINED, with sclk / power polled meanwhile:
# what an fp16 GEMM without any sign expansion reaches under the same power limit (beside tools/gemm_clock_probe.sh).
python - <<'PY' &
import time, torch
dev = torch.device("cuda:0")
T, K, N = 16384, 4096, 11008
x = torch.randn(T, K, device=dev).half(); w = torch.randn(N, K, device=dev).half(); y = torch.empty(T, N, device=dev, dtype=torch.float16)
for _ in range(5): torch.matmul(x, w.t(), out=y)
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(50): torch.matmul(x, w.t(), out=y)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print("dense fp16 torch.matmul [16384, 4096] x [4096, 11008]: %.3f ms per call, %.1f TFLOP/s sustained, %d calls" % (dt / n * 1e3, 2.0 * T * K * N * n / dt / 1e12, n), flush=True)
PY
sleep 2.5
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.4; done
wait
