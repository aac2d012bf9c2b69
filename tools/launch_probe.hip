// Launch-cost probe: how long does a dependent chain of tiny kernels take on this GPU, as a
// function of grid, block size, dynamic LDS, kernarg size and VGPR budget?  (hipcc tools/launch_probe.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct Big { long a[56]; };   // 448-byte by-value argument

__global__ void k_empty(int *p) { if (p == (int *)1) *p = 0; }
__global__ __launch_bounds__(512) void k_empty512(int *p) { if (p == (int *)1) *p = 0; }
__global__ __launch_bounds__(512) void k_big(const Big b) { if (b.a[0] == 12345) ((int *)b.a[1])[0] = 0; }
extern __shared__ char smem[];
__global__ __launch_bounds__(512) void k_lds(int *p) { if (p == (int *)1) smem[threadIdx.x] = 0; }
// many VGPRs: forces a large register allocation per wave
__global__ __launch_bounds__(512) void k_vgpr(float *p, int n)
{
    float r[100];
#pragma unroll
    for (int i = 0; i < 100; ++i) r[i] = p ? p[i] : (float)i * n;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 100; ++i) s += r[i] * r[(i * 7) % 100];
    if (n == 12345 && p) p[0] = s;
}

template <typename F> static float chain(F launch, int n, hipStream_t s)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / n;
}

template <typename F> static float graph_chain(F launch, int n, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / (5 * n);
}

int main()
{
    hipStream_t s; hipStreamCreate(&s);
    int *d; hipMalloc(&d, 1 << 20);
    Big b = {}; b.a[1] = (long)d;
    hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int N = 200;
#define RUN(name, expr)                                                                      \
    { auto f = [&]() { expr; };                                                              \
      float a = chain(f, N, s), g = graph_chain(f, N, s);                                    \
      printf("%-44s direct %7.2f us/launch   graph %7.2f us/launch\n", name, a, g); }
    RUN("empty  grid 1 x 64", hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, d));
    RUN("empty  grid 256 x 256", hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, d));
    RUN("empty  grid 256 x 512", hipLaunchKernelGGL(k_empty512, dim3(256), dim3(512), 0, s, d));
    RUN("empty  grid 1024 x 256", hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, d));
    RUN("bigarg grid 256 x 512 (448 B kernarg)", hipLaunchKernelGGL(k_big, dim3(256), dim3(512), 0, s, b));
    RUN("lds 10KB grid 256 x 512", hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 10 * 1024, s, d));
    RUN("lds 24KB grid 256 x 512", hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 24 * 1024, s, d));
    RUN("lds 48KB grid 256 x 512", hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 48 * 1024, s, d));
    RUN("lds 100KB grid 256 x 512", hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 100 * 1024, s, d));
    RUN("vgpr100 grid 256 x 512", hipLaunchKernelGGL(k_vgpr, dim3(256), dim3(512), 0, s, (float *)nullptr, 3));
    RUN("vgpr100 grid 256 x 512 + lds 24KB", hipLaunchKernelGGL(k_vgpr, dim3(256), dim3(512), 24 * 1024, s, (float *)nullptr, 3));
    return 0;
}
