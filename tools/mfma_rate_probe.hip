// MFMA rate probe (gfx950): cycles per matrix instruction seen by a wave, 1 / 2 waves per SIMD, four independent
// accumulators, alone and with the 4 v_and per MFMA that the decode GEMV's integer path issues.  Answers which
// shape / type gives the most K per matrix-pipe cycle for a batch-1 GEMV (few useful B columns).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_rate_probe tools/mfma_rate_probe.hip && tools/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int ANDS>
__global__ void k_rate(unsigned long long *out, int *sink, int n, unsigned seed)
{
    unsigned w[4] = {seed * (threadIdx.x + 1), seed ^ 0x9e3779b9u, seed + threadIdx.x, ~seed};
    i32x4 a4 = {1, 2, 3, 4}, b4 = {(int)threadIdx.x, 5, 6, 7};
    i32x8 a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f16x8 ah = {1, 2, 3, 4, 5, 6, 7, 8}, bh = {1, 1, 1, 1, 1, 1, 1, 1};
    i32x4 ci[4] = {};
    f32x4 cf[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (ANDS) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) { a4[v] = (int)(w[d] & (0x01010101u << ((r & 1) * 4 + v))); a8[v] = a4[v]; }
                }
                if (KIND == 0) ci[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, ci[d], 0, 0, 0);
                if (KIND == 1) cf[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, cf[d], 0, 0, 0);
                if (KIND == 2) cf[d] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, cf[d], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // fp8 x fp8
                if (KIND == 3) cf[d] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, cf[d], 4, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // fp4 (A) x fp8 (B)
                if (KIND == 4) cf[d] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, cf[d], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // fp4 x fp4
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) s += ci[d][0] + (int)cf[d][0];
    if (s == 0x12345) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND, int ANDS> static void run(const char *name, int threads, unsigned long long *d_out, int *sink)
{
    const int n = 32;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rate<KIND, ANDS>), dim3(256), dim3(threads), 0, 0, d_out, sink, n, 12345u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) v.push_back((double)h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    const double nm = (double)n * 8 * 4;
    printf("%-28s %s  %d waves/SIMD: %.1f cycles per MFMA per wave, %.1f per SIMD\n", name, ANDS ? "+4 v_and" : "bare    ", threads / 256,
           v[v.size() / 2] / nm, v[v.size() / 2] / nm / (threads / 256));
}
int main()
{
    unsigned long long *d; int *s; hipMalloc(&d, 256 * 16 * 8); hipMalloc(&s, 4096 * 4);
    for (int t : {256, 512}) {
        run<0, 0>("i32_16x16x64_i8", t, d, s); run<0, 1>("i32_16x16x64_i8", t, d, s);
        run<1, 0>("f32_16x16x32_f16", t, d, s);
        run<2, 0>("f32_16x16x128 fp8 x fp8", t, d, s); run<2, 1>("f32_16x16x128 fp8 x fp8", t, d, s);
        run<3, 0>("f32_16x16x128 fp4 x fp8", t, d, s); run<3, 1>("f32_16x16x128 fp4 x fp8", t, d, s);
        run<4, 0>("f32_16x16x128 fp4 x fp4", t, d, s);
    }
    return 0;
}
