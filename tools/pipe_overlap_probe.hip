// Probe: how the matrix pipe and the VALU of one gfx950 SIMD share time.  Every variant runs the same two instruction streams --
//   M: 16 x v_mfma_f32_16x16x32_f16 on 8 independent accumulators        V: 64 x v_fma_f32 on 8 independent chains
//   (E: 64 x v_exp_f32, P: 32 x v_pk_fma_f32 for the VALU rates the attention kernel's accounting uses)
// -- for R iterations, as volatile asm so that the stream is exactly what is written, and reports shader cycles per iteration
// (s_memtime, first wave of each group) with one workgroup per CU:
//   m1 / v1 / e1 / p1   one wave per SIMD, one stream
//   mv1                 one wave per SIMD, the two streams interleaved in the SAME wave (1 MFMA, 4 VALU, ...)
//   m2 / v2             two waves per SIMD, both the same stream
//   m+v                 two waves per SIMD: waves 0-3 run M, waves 4-7 run V  (the ping-pong arrangement of tools/attic/ob_flash_pp.h)
//   hipcc -O2 --offload-arch=gfx950 tools/pipe_overlap_probe.hip -o tools/pipe_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));
typedef float float2_ __attribute__((ext_vector_type(2)));

#define MF(c) "v_mfma_f32_16x16x32_f16 %" #c ", %8, %9, %" #c "\n\t"
#define FM(c) "v_fma_f32 %" #c ", %" #c ", %8, %9\n\t"
#define EX(c) "v_exp_f32 %" #c ", %" #c "\n\t"
#define PK(c) "v_pk_fma_f32 %" #c ", %" #c ", %8, %9\n\t"

struct St {
    float4_ c[8];
    float f[8];
    float2_ p[8];
    half8 a, b;
    float x, y;
    float2_ x2, y2;
};
__device__ __forceinline__ void stream_m(St &s)
{
    asm volatile(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7)
                 : "+v"(s.c[0]), "+v"(s.c[1]), "+v"(s.c[2]), "+v"(s.c[3]), "+v"(s.c[4]), "+v"(s.c[5]), "+v"(s.c[6]), "+v"(s.c[7]) : "v"(s.a), "v"(s.b));
}
#define V8 FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7)
__device__ __forceinline__ void stream_v(St &s)
{
    asm volatile(V8 V8 V8 V8 V8 V8 V8 V8
                 : "+v"(s.f[0]), "+v"(s.f[1]), "+v"(s.f[2]), "+v"(s.f[3]), "+v"(s.f[4]), "+v"(s.f[5]), "+v"(s.f[6]), "+v"(s.f[7]) : "v"(s.x), "v"(s.y));
}
#define E8 EX(0) EX(1) EX(2) EX(3) EX(4) EX(5) EX(6) EX(7)
__device__ __forceinline__ void stream_e(St &s)
{
    asm volatile(E8 E8 E8 E8 E8 E8 E8 E8
                 : "+v"(s.f[0]), "+v"(s.f[1]), "+v"(s.f[2]), "+v"(s.f[3]), "+v"(s.f[4]), "+v"(s.f[5]), "+v"(s.f[6]), "+v"(s.f[7]) : "v"(s.x), "v"(s.y));
}
#define P8 PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7)
__device__ __forceinline__ void stream_p(St &s)
{
    asm volatile(P8 P8 P8 P8
                 : "+v"(s.p[0]), "+v"(s.p[1]), "+v"(s.p[2]), "+v"(s.p[3]), "+v"(s.p[4]), "+v"(s.p[5]), "+v"(s.p[6]), "+v"(s.p[7]) : "v"(s.x2), "v"(s.y2));
}
// the two streams in one wave: MFMA k followed by VALU 4k .. 4k + 3 (16 MFMAs, 64 VALU)
#define MV(c, d) "v_mfma_f32_16x16x32_f16 %" #c ", %16, %17, %" #c "\n\tv_fma_f32 %" #d ", %" #d ", %18, %19\n\tv_fma_f32 %" #d ", %" #d ", %18, %19\n\t" \
                 "v_fma_f32 %" #d ", %" #d ", %18, %19\n\tv_fma_f32 %" #d ", %" #d ", %18, %19\n\t"
__device__ __forceinline__ void stream_mv(St &s)
{
    asm volatile(MV(0, 8) MV(1, 9) MV(2, 10) MV(3, 11) MV(4, 12) MV(5, 13) MV(6, 14) MV(7, 15) MV(0, 8) MV(1, 9) MV(2, 10) MV(3, 11) MV(4, 12) MV(5, 13) MV(6, 14) MV(7, 15)
                 : "+v"(s.c[0]), "+v"(s.c[1]), "+v"(s.c[2]), "+v"(s.c[3]), "+v"(s.c[4]), "+v"(s.c[5]), "+v"(s.c[6]), "+v"(s.c[7]),
                   "+v"(s.f[0]), "+v"(s.f[1]), "+v"(s.f[2]), "+v"(s.f[3]), "+v"(s.f[4]), "+v"(s.f[5]), "+v"(s.f[6]), "+v"(s.f[7])
                 : "v"(s.a), "v"(s.b), "v"(s.x), "v"(s.y));
}

// MFMA stream fed from LDS the way the attention kernel's score product is: one ds_read_b128 per two MFMAs, requested AHEAD fragments
// before its use (a ring of 8 fragment registers), counted waits.  MODE 5: AHEAD = 4, MODE 6: AHEAD = 7, MODE 7: the reads only.
template <int AHEAD, bool WITH_MFMA>
__device__ __forceinline__ void stream_mr(St &s, half8 (&kf)[8], unsigned lds_addr)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[(i + AHEAD) & 7]) : "v"(lds_addr), "n"(((i + AHEAD) & 7) * 1024));
        if (AHEAD == 4) asm volatile("s_waitcnt lgkmcnt(4)");
        else asm volatile("s_waitcnt lgkmcnt(7)");
        if (WITH_MFMA) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s.c[(2 * i) & 7]) : "v"(kf[i & 7]), "v"(s.b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s.c[(2 * i + 1) & 7]) : "v"(kf[i & 7]), "v"(s.b));
        }
    }
}

// mode: 0 M, 1 V, 2 E, 3 P, 4 MV (same wave); group B (waves 4-7 of a 512-thread workgroup) runs MB.  One loop per group and
// compile-time modes: a run-time choice inside the loop costs ~60 register copies per iteration at the merge points.
template <int MODE>
__device__ __forceinline__ unsigned long long run(St &s, int R)
{
    half8 kf[8];
    for (int i = 0; i < 8; ++i) kf[i] = s.a;
    const unsigned lds_addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
    if (MODE >= 5) for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[i]) : "v"(lds_addr), "n"(i * 1024));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < R; ++r) {
        if (MODE == 0) stream_m(s);
        else if (MODE == 1) stream_v(s);
        else if (MODE == 2) stream_e(s);
        else if (MODE == 3) stream_p(s);
        else if (MODE == 4) stream_mv(s);
        else if (MODE == 5) stream_mr<4, true>(s, kf, lds_addr);
        else if (MODE == 6) stream_mr<7, true>(s, kf, lds_addr);
        else stream_mr<4, false>(s, kf, lds_addr);
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (MODE >= 5) for (int i = 0; i < 8; ++i) s.f[i] += (float)kf[i][0];
    asm volatile("s_nop 15\n\ts_nop 15");
    return __builtin_amdgcn_s_memtime() - t0;
}
template <int MA, int MB>
__global__ __launch_bounds__(1024) void probe(unsigned long long *out, int R)
{
    extern __shared__ char pad[];       // the launch asks for > half the LDS: one workgroup per CU
    St s;
    for (int i = 0; i < 8; ++i) { s.c[i] = (float4_){0.f, 0.f, 0.f, 0.f}; s.f[i] = 0.001f * threadIdx.x + i; s.p[i] = (float2_){s.f[i], -s.f[i]}; }
    for (int i = 0; i < 8; ++i) { s.a[i] = (_Float16)(0.01f * (threadIdx.x & 7)); s.b[i] = (_Float16)0.5f; }
    s.x = 0.999f; s.y = 0.001f; s.x2 = (float2_){0.999f, 0.998f}; s.y2 = (float2_){0.001f, 0.002f};
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    unsigned long long dt;
    if (wave < 4) dt = run<MA>(s, R);
    else dt = run<MB>(s, R);
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += s.c[i][0] + s.f[i] + s.p[i][0];
    if (acc == 12345.678f) out[1023] = 1;     // keeps the results alive
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = dt;     // (waves 8-15 of a 1024-thread workgroup run MB too)
}
typedef void (*probe_fn)(unsigned long long *, int);

int main()
{
    unsigned long long *d, h[16];
    hipMalloc(&d, 8192);
    const int R = 40000;
    struct { const char *name; int threads; probe_fn fn; } v[] = {
        {"m1   one wave/SIMD, MFMA stream (16 MFMA)", 256, probe<0, 0>}, {"v1   one wave/SIMD, VALU stream (64 v_fma_f32)", 256, probe<1, 1>},
        {"e1   one wave/SIMD, 64 v_exp_f32", 256, probe<2, 2>}, {"p1   one wave/SIMD, 32 v_pk_fma_f32 (= 64 fma)", 256, probe<3, 3>},
        {"mv1  one wave/SIMD, both streams interleaved in the same wave", 256, probe<4, 4>},
        {"m2   two waves/SIMD, both the MFMA stream", 512, probe<0, 0>}, {"v2   two waves/SIMD, both the VALU stream", 512, probe<1, 1>},
        {"m+v  two waves/SIMD, waves 0-3 MFMA stream, waves 4-7 VALU stream", 512, probe<0, 1>},
        {"m+e  two waves/SIMD, waves 0-3 MFMA stream, waves 4-7 v_exp stream", 512, probe<0, 2>},
        {"v+m  two waves/SIMD, waves 0-3 VALU stream, waves 4-7 MFMA stream", 512, probe<1, 0>},
        {"e+m  two waves/SIMD, waves 0-3 v_exp stream, waves 4-7 MFMA stream", 512, probe<2, 0>},
        {"m+p  two waves/SIMD, waves 0-3 MFMA stream, waves 4-7 v_pk_fma stream", 512, probe<0, 3>},
        {"mv+v two waves/SIMD, waves 0-3 interleaved, waves 4-7 VALU stream", 512, probe<4, 1>},
        {"mv2  two waves/SIMD, both interleaved streams", 512, probe<4, 4>},
        {"v4   FOUR waves/SIMD, all the VALU stream (slowest of waves 0-3 | 4-7 shown)", 1024, probe<1, 1>},
        {"e4   FOUR waves/SIMD, all the v_exp stream", 1024, probe<2, 2>},
        {"p4   FOUR waves/SIMD, all the v_pk_fma stream", 1024, probe<3, 3>},
        {"p2   two waves/SIMD, both the v_pk_fma stream", 512, probe<3, 3>},
        {"e2   two waves/SIMD, both the v_exp stream", 512, probe<2, 2>},
        {"mr1  one wave/SIMD, 16 MFMA + 8 ds_read_b128, fragments 4 ahead", 256, probe<5, 5>},
        {"mr1' one wave/SIMD, 16 MFMA + 8 ds_read_b128, fragments 7 ahead", 256, probe<6, 6>},
        {"r1   one wave/SIMD, the 8 ds_read_b128 alone (4 ahead)", 256, probe<7, 7>},
        {"mr2  two waves/SIMD, both 16 MFMA + 8 ds_read_b128 (4 ahead)", 512, probe<5, 5>},
        {"mr+v two waves/SIMD, waves 0-3 MFMA + reads, waves 4-7 VALU stream", 512, probe<5, 1>},
        {"mr+mv two waves/SIMD, waves 0-3 MFMA + reads, waves 4-7 interleaved MFMA + VALU", 512, probe<5, 4>}};
    printf("cycles per iteration (first wave of waves 0-3 | of waves 4-7); an iteration = 16 MFMAs (256 pipe cycles) and / or 64 VALU ops\n");
    for (auto &x : v) {
        hipMemset(d, 0, 128);
        hipFuncSetAttribute((const void *)x.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 100000);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(x.fn, dim3(256), dim3(x.threads), 100000, 0, d, R);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(x.fn, dim3(256), dim3(x.threads), 100000, 0, d, R);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        unsigned long long tmax = 0;
        for (int w = 0; w < x.threads / 64; ++w) tmax = h[w] > tmax ? h[w] : tmax;
        if (x.threads == 1024) { h[0] = 0; for (int w = 0; w < 16; ++w) h[0] = h[w] > h[0] ? h[w] : h[0]; h[4] = h[0]; }
        if (x.threads == 256) printf("%-72s %7.1f            ", x.name, (double)h[0] / R);
        else printf("%-72s %7.1f | %7.1f  ", x.name, (double)h[0] / R, (double)h[4] / R);
        printf("  kernel %.2f ms: %.0f s_memtime ticks per microsecond\n", ms, tmax / (ms * 1e3));
    }
    return 0;
}
