// Probe 3: straight-line code executed once (instruction-fetch bound?) vs the same work in a loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N, bool LOOP>
__global__ __launch_bounds__(512) void k_work(float *p, int n)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i);
    if (LOOP) {
#pragma unroll 1
        for (int it = 0; it < N / 64; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = a[i] * 1.0001f + (float)n;
        }
    } else {
#pragma unroll
        for (int it = 0; it < N / 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i] * (1.0001f + 0.0001f * (it & 7)) + (float)n;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    p[blockIdx.x * 512 + threadIdx.x] = s;
}
template <typename F> static float graph_chain(F launch, int n, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / (5 * n);
}
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    float *d; hipMalloc(&d, 1 << 20);
#define RUN(N, L) { auto f = [&]() { hipLaunchKernelGGL((k_work<N, L>), dim3(256), dim3(512), 0, s, d, 3); }; \
    printf("%6d fma/lane %-13s: %7.2f us/launch\n", N, L ? "(loop of 64)" : "(straight)", graph_chain(f, 100, s)); }
    RUN(512, true) RUN(512, false) RUN(2048, true) RUN(2048, false) RUN(4096, true) RUN(4096, false) RUN(8192, true) RUN(8192, false)
    return 0;
}
