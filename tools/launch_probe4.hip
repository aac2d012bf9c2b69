// Probe 4: cost of the kernarg fetch at the head of a kernel, and whether kernarg preload
// (-mllvm -amdgpu-kernarg-preload-count=N, scalar arguments) hides it.
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { const float *p[16]; int n[16]; };
__global__ __launch_bounds__(512) void k_empty(float *o, const float *a, int n) {}
// one dependent chain: kernarg -> global load -> store
__global__ __launch_bounds__(512) void k_args(float *o, const float *a, int n)
{
    const float v = a[threadIdx.x + n];
    if (v == 12345.f) o[blockIdx.x * 512 + threadIdx.x] = v;
}
__global__ __launch_bounds__(512) void k_struct(const Big b)
{
    const float v = b.p[15][threadIdx.x + b.n[15]];
    if (v == 12345.f) const_cast<float *>(b.p[0])[blockIdx.x * 512 + threadIdx.x] = v;
}
// the load address does not depend on the kernarg contents at all (only the condition does)
__global__ __launch_bounds__(512) void k_noarg(float *o, const float *a, int n)
{
    if (n == 77) o[threadIdx.x] = 1.f;
}
// spin for `ticks` of the 100 MHz constant clock; wave 0 of block 0 also records the s_memtime delta
__global__ __launch_bounds__(512) void k_spin(unsigned long long *o, int ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(1);
    if (blockIdx.x == 0 && threadIdx.x == 0) { o[0] = __builtin_readcyclecounter() - c0; o[1] = __builtin_amdgcn_s_memrealtime() - t0; }
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
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / (10 * n);
}
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    float *d, *a; hipMalloc(&d, 1 << 20); hipMalloc(&a, 1 << 20); hipMemset(a, 0, 1 << 20);
    Big b = {}; for (int i = 0; i < 16; ++i) { b.p[i] = i ? a : d; b.n[i] = 3; }
    auto f0 = [&]() { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s, d, a, 3); };
    auto f1 = [&]() { hipLaunchKernelGGL(k_args, dim3(256), dim3(512), 0, s, d, a, 3); };
    auto f2 = [&]() { hipLaunchKernelGGL(k_struct, dim3(256), dim3(512), 0, s, b); };
    auto f3 = [&]() { hipLaunchKernelGGL(k_noarg, dim3(256), dim3(512), 0, s, d, a, 3); };
    printf("empty            : %6.2f us/launch\n", graph_chain(f0, 200, s));
    printf("kernarg test only: %6.2f us/launch\n", graph_chain(f3, 200, s));
    printf("args -> load     : %6.2f us/launch\n", graph_chain(f1, 200, s));
    printf("struct -> load   : %6.2f us/launch\n", graph_chain(f2, 200, s));
    unsigned long long *t; hipMalloc(&t, 64);
    for (int ticks : {0, 100, 200, 400, 800}) {
        auto fs = [&]() { hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 0, s, t, ticks); };
        const float us = graph_chain(fs, 100, s);
        unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("spin %4.1f us: %6.2f us/launch (overhead %5.2f)  s_memtime delta %llu over %llu ticks -> %.1f MHz\n", ticks / 100.0, us, us - ticks / 100.0, h[0], h[1], h[1] ? h[0] * 100.0 / h[1] : 0.0);
    }
    return 0;
}
