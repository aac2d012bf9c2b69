// Row-statistics helpers shared by the decode kernels (ob_decode.h) and the key-block attention (ob_fdec.h): the LayerNorm element
// form and the combine of a producer's per-16-row-tile partials (see ob_decode.h, "Producer-side LayerNorm partials").
#pragma once
#include "ob_common.h"

__device__ __forceinline__ float ob_ln_apply(float u, float mean, float rstd)
{
    return ob_round_h((u - mean) * rstd);
}

// The same for a vector whose length is a runtime value (<= 16384): blocks of 256 tiles beyond
// n / 16 are neither loaded nor counted.
struct ObTileStatsRt { ob_float4 a[2]; };          // block 0 (the first 4096 elements); further blocks are re-read in the combine
__device__ __forceinline__ void ob_tiles_load_rt(ObTileStatsRt &r, const float *st, int n, int lane)
{
    const ob_float4 *p = reinterpret_cast<const ob_float4 *>(st + (size_t)lane * 8);
    r.a[0] = p[0];
    r.a[1] = p[1];
}
__device__ __forceinline__ void ob_tiles_combine_rt(const ObTileStatsRt &r, const float *st, int n, float eps, int lane, float &mean, float &rstd)
{
    const int ntiles = n >> 4;
    float s = 0.f;
    for (int v = 0; v * 256 < ntiles; ++v) {               // uniform trip count; 1 for vectors up to 4096
        ob_float4 a0 = r.a[0], a1 = r.a[1];
        if (v) { const ob_float4 *p = reinterpret_cast<const ob_float4 *>(st + (size_t)(v * 64 + lane) * 8); a0 = p[0]; a1 = p[1]; }
        const int t = (v * 64 + lane) * 4;
        s += (t < ntiles ? a0[0] : 0.f) + (t + 1 < ntiles ? a0[2] : 0.f) + (t + 2 < ntiles ? a1[0] : 0.f) + (t + 3 < ntiles ? a1[2] : 0.f);
    }
    s = ob_wave_sum(s);
    const float inv_n = __builtin_amdgcn_rcpf((float)n);
    mean = s * inv_n;
    float m2 = 0.f;
    for (int v = 0; v * 256 < ntiles; ++v) {
        ob_float4 a0 = r.a[0], a1 = r.a[1];
        if (v) { const ob_float4 *p = reinterpret_cast<const ob_float4 *>(st + (size_t)(v * 64 + lane) * 8); a0 = p[0]; a1 = p[1]; }
        const int t = (v * 64 + lane) * 4;
        const float sv[4] = {a0[0], a0[2], a1[0], a1[2]}, qv[4] = {a0[1], a0[3], a1[1], a1[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = __builtin_fmaf(sv[i], 0.0625f, -mean);
            m2 += t + i < ntiles ? __builtin_fmaf(16.0f * d, d, qv[i]) : 0.f;
        }
    }
    m2 = ob_wave_sum(m2);
    rstd = __builtin_amdgcn_rsqf(fmaxf(m2 * inv_n, 0.f) + eps);
}
