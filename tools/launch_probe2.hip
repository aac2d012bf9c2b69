// Probe 2: does an early-returning kernel's launch cost depend on its VGPR allocation / code size?
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NV>
__global__ __launch_bounds__(512) void k_vg(int *p)
{
    if (p != (int *)1) return;
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = (float)p[i + threadIdx.x];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i] * r[(i * 7 + 3) % NV];
    p[threadIdx.x] = (int)s;
}

template <int NC>
__global__ __launch_bounds__(512) void k_code(int *p)
{
    if (p != (int *)1) return;
    float s = (float)threadIdx.x;
#pragma unroll
    for (int i = 0; i < NC; ++i) s = s * 1.0001f + (float)i;
    p[threadIdx.x] = (int)s;
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
    const int N = 200;
#define RUN(name, K, G, B, L)                                                                \
    { auto f = [&]() { hipLaunchKernelGGL(K, dim3(G), dim3(B), L, s, d); };                  \
      printf("%-40s grid %4d x %3d lds %6d: graph %7.2f us/launch\n", name, G, B, L, graph_chain(f, N, s)); }
    RUN("vg<8>", k_vg<8>, 256, 512, 0);
    RUN("vg<32>", k_vg<32>, 256, 512, 0);
    RUN("vg<64>", k_vg<64>, 256, 512, 0);
    RUN("vg<100>", k_vg<100>, 256, 512, 0);
    RUN("vg<100>", k_vg<100>, 256, 256, 0);
    RUN("vg<100>", k_vg<100>, 256, 64, 0);
    RUN("vg<100>", k_vg<100>, 64, 512, 0);
    RUN("vg<100>", k_vg<100>, 1024, 512, 0);
    RUN("vg<100>", k_vg<100>, 256, 512, 20480);
    RUN("code<100>", k_code<100>, 256, 512, 0);
    RUN("code<1000>", k_code<1000>, 256, 512, 0);
    RUN("code<3000>", k_code<3000>, 256, 512, 0);
    RUN("code<10000>", k_code<10000>, 256, 512, 0);
    return 0;
}
