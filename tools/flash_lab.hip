// Development bench for the prefill attention kernel, without torch: random fp16 q / k / v at the BASELINE config-3 shape,
// timed with HIP events, sampled query rows checked against an fp32 host restatement of modeling_bitllama.py:546-563, and
// (built with -DOB_FL_TRACE) an s_memtime timeline of two workgroups.  Variants are compile-time switches of ob_flash.h:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ionebit_amd/csrc -Itools [-DOB_FL_...] tools/flash_lab.hip -o tools/flash_lab
//   tools/flash_lab [B S H reps label]          (FL_DYN=40000 in the environment: one workgroup per CU)
// -DOB_FL_ABL=1|2|3: no softmax arithmetic / no staging / neither (timing only); -DOB_FL_DEFER_THR=0.0f: rescale on every new maximum;
// (the -DFL_K64 / -DFL_PP builds of rounds 3-4 ran the rejected rearrangements that lived under tools/attic: removed in round 5,
//  results in docs/experiments.md "Prefill attention, round 4", code in the git history)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "ob_flash.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline float rnd_uniform()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) + 1) / 16777217.0f;
}
static inline float rnd_normal() { return sqrtf(-2.0f * logf(rnd_uniform())) * cosf(6.2831853f * rnd_uniform()); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, S = argc > 2 ? atoi(argv[2]) : 2048, H = argc > 3 ? atoi(argv[3]) : 32;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    constexpr int D = 128;
    const size_t n = (size_t)B * S * H * D;
    std::vector<_Float16> q(n), k(n), v(n), o(n);
    for (size_t i = 0; i < n; ++i) { q[i] = (_Float16)rnd_normal(); k[i] = (_Float16)rnd_normal(); v[i] = (_Float16)rnd_normal(); }
    _Float16 *dq, *dk, *dv, *d_o;
    CK(hipMalloc(&dq, 2 * n)); CK(hipMalloc(&dk, 2 * n)); CK(hipMalloc(&dv, 2 * n)); CK(hipMalloc(&d_o, 2 * n));
    CK(hipMemcpy(dq, q.data(), 2 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, k.data(), 2 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, v.data(), 2 * n, hipMemcpyHostToDevice));
    CK(hipMemset(d_o, 0, 2 * n));
#define FL_THREADS OB_FL_THREADS
#define FL_KERNEL ob_flash_fwd_kernel<D>
    const int nmb = (S + OB_FL_BM - 1) / OB_FL_BM;
    const int dyn0 = 0;
    ObFlashArgs a = {dq, dk, dv, d_o, nullptr, S, H, H, S, 0, 1.4426950408889634f / sqrtf((float)D), nmb};
#ifdef OB_FL_TRACE
    unsigned long long *dtr;
    const size_t ntr = 2 * (FL_THREADS / 64) * 66 * 8;
    CK(hipMalloc(&dtr, 8 * ntr)); CK(hipMemset(dtr, 0, 8 * ntr));
    a.trace = dtr;
#endif
    const dim3 grid((unsigned)(((nmb + 1) / 2) * H * B));
    const int dyn = dyn0 + (getenv("FL_DYN") ? atoi(getenv("FL_DYN")) : 0);       // unused dynamic LDS: FL_DYN=40000 leaves room for ONE workgroup per CU
    if (dyn) CK(hipFuncSetAttribute((const void *)FL_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((FL_KERNEL), grid, dim3(FL_THREADS), dyn, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((FL_KERNEL), grid, dim3(FL_THREADS), dyn, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double flop = 4.0 * B * H * (double)S * S * D / 2;
    CK(hipMemcpy(o.data(), d_o, 2 * n, hipMemcpyDeviceToHost));

    // sampled rows against fp32
    double max_err = 0, max_ref = 0;
    for (int t = 0; t < 96; ++t) {
        const int b = t % B, h = (t * 7) % H, s = t < 8 ? t : (t < 16 ? S - 1 - (t - 8) : (int)(rnd_uniform() * (S - 1)));
        const _Float16 *qr = &q[(((size_t)b * S + s) * H + h) * D];
        std::vector<float> p(s + 1);
        float mx = -INFINITY;
        for (int j = 0; j <= s; ++j) {
            const _Float16 *kr = &k[(((size_t)b * H + h) * S + j) * D];
            float acc = 0;
            for (int d = 0; d < D; ++d) acc += (float)qr[d] * (float)kr[d];
            p[j] = acc / sqrtf((float)D); mx = fmaxf(mx, p[j]);
        }
        double l = 0;
        for (int j = 0; j <= s; ++j) { p[j] = expf(p[j] - mx); l += p[j]; }
        for (int d = 0; d < D; ++d) {
            double acc = 0;
            for (int j = 0; j <= s; ++j) acc += (double)p[j] * (float)v[(((size_t)b * H + h) * S + j) * D + d];
            const double ref = acc / l, got = (float)o[(((size_t)b * S + s) * H + h) * D + d];
            max_err = fmax(max_err, fabs(ref - got)); max_ref = fmax(max_ref, fabs(ref));
        }
    }
    printf("%-28s %.3f ms  %.0f TFLOP/s   max |err| %.2e (max |ref| %.2f)%s\n", argc > 5 ? argv[5] : "flash", ms, flop / ms / 1e9, max_err, max_ref,
           max_err < 4e-3 ? "" : "   <-- WRONG");
#ifdef OB_FL_TRACE
    std::vector<unsigned long long> tr(ntr);
    CK(hipMemcpy(tr.data(), dtr, 8 * ntr, hipMemcpyDeviceToHost));
    const int NW = FL_THREADS / 64;
    for (int slot = 0; slot < 2; ++slot) {
        printf("workgroup slot %d: SIMD of waves 0 .. %d:", slot, NW - 1);
        for (int w = 0; w < NW; ++w) printf(" %llu", (tr[((size_t)slot * NW + w) * 66 * 8 + 65 * 8 + 4] >> 4) & 3);
        printf("\n");
    }
    for (int slot = 0; slot < 2; ++slot)
        for (int w = 0; w < NW; w += NW - 1) {
            printf("trace: workgroup slot %d wave %d -- per key block, cycles since the block's first stamp\n", slot, w);
            const unsigned long long *t = &tr[((size_t)slot * NW + w) * 66 * 8];
            for (int ps = 0; ps < 2; ++ps)
                printf("  pass %d: prologue %llu  key loop %llu  epilogue %llu   (next pass starts +%llu)\n", ps, t[(64 + ps) * 8 + 1] - t[(64 + ps) * 8], t[(64 + ps) * 8 + 2] - t[(64 + ps) * 8 + 1],
                       t[(64 + ps) * 8 + 3] - t[(64 + ps) * 8 + 2], ps == 0 ? t[65 * 8] - t[64 * 8 + 3] : 0ull);
            unsigned long long prev0 = 0;
            for (int kb = 0; kb < 64; ++kb) {
                if (!t[kb * 8]) continue;
                printf("  kb %2d  (+%5llu since previous)", kb, prev0 ? t[kb * 8] - prev0 : 0ull);
                for (int i = 1; i < 8 && t[kb * 8 + i]; ++i) printf("  %5llu", t[kb * 8 + i] - t[kb * 8]);
                printf("\n");
                prev0 = t[kb * 8];
            }
        }
#endif
    return 0;
}
