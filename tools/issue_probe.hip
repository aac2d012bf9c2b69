// Issue-rate probe: cycles per instruction seen by ONE wave when W waves per SIMD run the same
// instruction stream (256 workgroups, one per CU).  DEP = 1: one dependent chain; DEP = 4: four
// independent chains interleaved (ILP); mix of full-rate VALU ops as in the decode prologue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
template <int DEP>
__global__ void k_issue(unsigned long long *out, float *sink, int n)
{
    float a[4] = {1.0f + threadIdx.x, 2.0f, 3.0f, 4.0f};
    unsigned u[4] = {threadIdx.x, 7u, 9u, 11u};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int d = 0; d < DEP; ++d) {
                a[d] = __builtin_fmaf(a[d], 1.0001f, 0.5f);
                u[d] = __builtin_amdgcn_perm(u[d], __float_as_uint(a[d]), 0x05010400u);
                u[d] = (u[d] + 0x00808080u) ^ 0x00808080u;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; unsigned x = 0;
#pragma unroll
    for (int d = 0; d < DEP; ++d) { s += a[d]; x ^= u[d]; }
    if (s == 12345.f && x == 77u) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int DEP> static void run(int threads, unsigned long long *d_out, float *sink)
{
    const int n = 64;
    hipLaunchKernelGGL(k_issue<DEP>, dim3(256), dim3(threads), 0, 0, d_out, sink, n);
    hipLaunchKernelGGL(k_issue<DEP>, dim3(256), dim3(threads), 0, 0, d_out, sink, n);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) v.push_back((double)h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    const double instr = (double)n * 8 * DEP * 4;      // fma, perm, add, xor per chain step
    printf("threads %4d (%d waves/SIMD) chains %d: %.0f cycles for %.0f instr -> %.2f cycles/instr/wave, %.2f cycles/instr/SIMD\n",
           threads, threads / 256, DEP, v[v.size() / 2], instr, v[v.size() / 2] / instr, v[v.size() / 2] / instr / (threads / 256));
}
int main()
{
    unsigned long long *d; float *s; hipMalloc(&d, 256 * 16 * 8); hipMalloc(&s, 4096);
    for (int t : {256, 512, 1024}) { run<1>(t, d, s); run<2>(t, d, s); run<4>(t, d, s); }
    return 0;
}
