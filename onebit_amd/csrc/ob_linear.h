// OneBit 1-bit linear layer kernels for gfx950 -- first (v1) generation.
//
// ob_mm16_f16_kernel: z[n][t] = sum_k s[n][k] * fp16(x[t][k] * h[k]) for a 16-row x 16-token
// tile per workgroup on v_mfma_f32_16x16x32_f16.  The packed sign words are the MFMA A operand
// (expanded to +-1.0 fp16 in registers, never through memory), the scaled activations the
// B operand, fp32 accumulation -- the arithmetic of bitnet.py:113-115.  K is split across the
// workgroup's waves and reduced through LDS; the epilogue applies the reference's fp16
// rounding of z and the *g of bitnet.py:116.
#pragma once
#include "ob_common.h"

// One packed dword = 32 weights of this lane's row, k = kb .. kb+31.  Four MFMA sub-steps of
// 8 k each; the lane supplies A[row][8 k] (expanded signs) and B[8 k][token] = fp16(x*h).
__device__ __forceinline__ void ob_mm16_dword(uint32_t w, int kb, bool bvalid,
                                              const _Float16 *__restrict__ xrow,
                                              const _Float16 *__restrict__ h, ob_float4 &acc)
{
    ob_half8 bop[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (bvalid) {
            const ob_half8 xv = *reinterpret_cast<const ob_half8 *>(xrow + kb + 8 * s);
            const ob_half8 hv = *reinterpret_cast<const ob_half8 *>(h + kb + 8 * s);
            bop[s] = xv * hv;                      // v_pk_mul_f16: fp16(x*h), round-to-nearest-even
        } else {
            bop[s] = (ob_half8)(_Float16)0;
        }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        uint32_t e[8];
        ob_expand16((w >> (16 * hf)) & 0xffffu, e);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            ob_u32x4 av = {e[4 * s2 + 0], e[4 * s2 + 1], e[4 * s2 + 2], e[4 * s2 + 3]};
            ob_half8 aop;
            __builtin_memcpy(&aop, &av, 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bop[2 * hf + s2], acc, 0, 0, 0);
        }
    }
}

// grid = (ceil(N/16), ceil(T/16)), block = WAVES*64.
// PARTIAL: write raw fp32 sums to zp[T,N] (K-sharded path); else u = fp16(fp16(z)*g) to u[T,N].
template <int WAVES, bool PARTIAL>
__global__ __launch_bounds__(WAVES * 64) void ob_mm16_f16_kernel(
    const uint32_t *__restrict__ W, int64_t ldw_words, const _Float16 *__restrict__ x, int64_t ldx,
    const _Float16 *__restrict__ h, const _Float16 *__restrict__ g, _Float16 *__restrict__ u,
    float *__restrict__ zp, int T, int K, int N, int fast)
{
    __shared__ ob_float4 red[WAVES > 1 ? WAVES : 1][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, gq = lane >> 4;
    const int n0 = blockIdx.x * 16, t0 = blockIdx.y * 16;
    const int row = min(n0 + r, N - 1);
    const int tcol = t0 + r;
    const bool tvalid = tcol < T;
    const uint32_t *Wrow = W + (int64_t)row * ldw_words;
    const _Float16 *xrow = x + (int64_t)min(tcol, T - 1) * ldx;

    ob_float4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nsteps = fast ? (K >> 9) : 0;           // 512 k per dwordx4 step
    for (int step = wave; step < nsteps; step += WAVES) {
        const ob_u32x4 w4 = *reinterpret_cast<const ob_u32x4 *>(Wrow + step * 16 + gq * 4);
        const int kb0 = step * 512 + gq * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) ob_mm16_dword(w4[q], kb0 + 32 * q, tvalid, xrow, h, acc);
    }
    const int kdone = nsteps << 9;
    const int nblk = (K - kdone + 127) >> 7;           // 128 k per single-dword block
    for (int blk = wave; blk < nblk; blk += WAVES) {
        const int kb = kdone + blk * 128 + gq * 32;
        const bool kvalid = kb < K;
        const uint32_t w = kvalid ? Wrow[kb >> 5] : 0u;
        ob_mm16_dword(w, kvalid ? kb : 0, tvalid && kvalid, xrow, h, acc);
    }

    if (WAVES > 1) {
        red[wave][lane] = acc;
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w2 = 1; w2 < WAVES; ++w2) acc += red[w2][lane];
    }
    // D layout: lane holds rows n0 + 4*gq + i (i = 0..3) of token t0 + (lane & 15)
    if (!tvalid) return;
    const int nb = n0 + 4 * gq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = nb + i;
        if (n >= N) break;
        if (PARTIAL) {
            zp[(int64_t)tcol * N + n] = acc[i];
        } else {
            const float z16 = ob_round_h(acc[i]);
            u[(int64_t)tcol * N + n] = (_Float16)(z16 * (float)g[n]);
        }
    }
}

// Generic compatibility path: any K % 8 == 0, any row pitch, fp32 or fp16 activations (checkpoint
// loaded as fp32; odd shapes such as a 688-wide test config).  One wave per
// (row, token), one packed byte per lane per iteration, sign applied by conditional negate, fp32
// accumulate.  Not a performance path.
template <typename TX>
__global__ __launch_bounds__(64) void ob_simple_kernel(
    const uint8_t *__restrict__ W, int64_t ldw_bytes, const TX *__restrict__ x, int64_t ldx,
    const TX *__restrict__ h, float *__restrict__ zp, int T, int K, int N)
{
    const int n = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const uint8_t *Wrow = W + (int64_t)n * ldw_bytes;
    const TX *xr = x + (int64_t)t * ldx;
    float acc = 0.f;
    for (int j = lane; j < (K >> 3); j += 64) {
        const unsigned w = Wrow[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = (float)xr[8 * j + i] * (float)h[8 * j + i];
            if (sizeof(TX) == 2) a = ob_round_h(a);
            acc += ((w >> i) & 1u) ? -a : a;
        }
    }
    acc = ob_wave_sum(acc);
    if (lane == 0) zp[(int64_t)t * N + n] = acc;
}

// LayerNorm epilogue, one 256-thread workgroup per token.  Three passes over the (L2-resident)
// row: mean, centred variance, normalise -- the biased-variance / eps form of
// nn.LayerNorm(elementwise_affine=False) (bitnet.py:86,118), then the optional bias (:119-120).
//   FROM_Z: input is fp32 z (pre-g); u = g * z with the dtype's rounding points.
//   else   : input is u already (TD), normalised in place or into y.
template <typename TD, bool FROM_Z>
__global__ __launch_bounds__(256) void ob_layernorm_kernel(
    const float *z, const TD *uin, const TD *__restrict__ g, const TD *__restrict__ bias, TD *y,
    TD *uout, int N, float eps, int skip_ln)   // y may alias uin (in-place), so no restrict there
{
    __shared__ float red[8];
    const int64_t t = blockIdx.x;
    const float *zr = FROM_Z ? z + t * N : nullptr;
    const TD *ur = FROM_Z ? nullptr : uin + t * N;
    auto load_u = [&](int n) -> float {
        if (FROM_Z) {
            if (sizeof(TD) == 2) return ob_round_h(ob_round_h(zr[n]) * (float)g[n]);
            return zr[n] * (float)g[n];
        }
        return (float)ur[n];
    };
    if (skip_ln) {
        for (int n = threadIdx.x; n < N; n += 256) {
            const float v = load_u(n);
            y[t * N + n] = (TD)v;
            if (uout) uout[t * N + n] = (TD)v;
        }
        return;
    }
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) s += load_u(n);
    const float mean = ob_block_sum(s, red) / (float)N;
    float v = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float d = load_u(n) - mean;
        v += d * d;
    }
    const float var = ob_block_sum(v, red) / (float)N;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int n = threadIdx.x; n < N; n += 256) {
        const float uu = load_u(n);
        float o = (uu - mean) * rstd;
        if (sizeof(TD) == 2) o = ob_round_h(o);
        if (bias) {
            o += (float)bias[n];
            if (sizeof(TD) == 2) o = ob_round_h(o);
        }
        if (uout) uout[t * N + n] = (TD)uu;
        y[t * N + n] = (TD)o;
    }
}

// LayerNorm epilogue v2 for fp16 rows up to 16384 wide with N % 8 == 0: one 256-thread workgroup per
// token keeps the whole row in registers (8 x 16-byte loads per thread at most) -- one read, exact
// two-pass statistics (mean, then centred sum of squares), one write.  Same arithmetic as
// ob_layernorm_kernel; used whenever the shape allows.
template <bool FROM_Z>
__global__ __launch_bounds__(256) void ob_layernorm_rows_kernel(
    const float *z, const _Float16 *uin, const _Float16 *__restrict__ g, const _Float16 *__restrict__ bias,
    _Float16 *y, _Float16 *uout, int N, float eps, int skip_ln)
{
    __shared__ float red[16];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    float u[8][8];
    bool ok[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int base = (v * 256 + tid) * 8;
        ok[v] = base < N;
        if (ok[v]) {
            if (FROM_Z) {
                const ob_float4 a = *reinterpret_cast<const ob_float4 *>(z + t * N + base);
                const ob_float4 b = *reinterpret_cast<const ob_float4 *>(z + t * N + base + 4);
                const ob_half8 gv = *reinterpret_cast<const ob_half8 *>(g + base);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u[v][i] = ob_round_h(ob_round_h(a[i]) * (float)gv[i]);
                    u[v][i + 4] = ob_round_h(ob_round_h(b[i]) * (float)gv[i + 4]);
                }
            } else {
                const ob_half8 a = *reinterpret_cast<const ob_half8 *>(uin + t * N + base);
#pragma unroll
                for (int i = 0; i < 8; ++i) u[v][i] = (float)a[i];
            }
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (!skip_ln) {
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (ok[v]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s += u[v][i];
            }
        mean = ob_block_sum(s, red) / (float)N;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (ok[v]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = u[v][i] - mean; q = fmaf(d, d, q); }
            }
        const float var = ob_block_sum(q, red + 8) / (float)N;
        rstd = 1.0f / sqrtf(var + eps);
    }
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int base = (v * 256 + tid) * 8;
        if (ok[v]) {
            ob_half8 o, uo;
            ob_half8 bv = (ob_half8)(_Float16)0;
            if (bias && !skip_ln) bv = *reinterpret_cast<const ob_half8 *>(bias + base);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uo[i] = (_Float16)u[v][i];
                float r = skip_ln ? u[v][i] : ob_round_h((u[v][i] - mean) * rstd);
                if (bias && !skip_ln) r = ob_round_h(r + (float)bv[i]);
                o[i] = (_Float16)r;
            }
            *reinterpret_cast<ob_half8 *>(y + t * N + base) = o;
            if (uout) *reinterpret_cast<ob_half8 *>(uout + t * N + base) = uo;
        }
    }
}

// N-sharded LayerNorm halves (include/onebit.h): one 256-thread workgroup per token row slice.
template <typename TD>
__global__ __launch_bounds__(256) void ob_row_stats_kernel(const TD *__restrict__ u, float *__restrict__ stats, int n)
{
    __shared__ float red[8];
    const TD *row = u + (int64_t)blockIdx.x * n;
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) s += (float)row[j];
    const float mean = ob_block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) { const float d = (float)row[j] - mean; q += d * d; }
    q = ob_block_sum(q, red);
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = q; }
}

// Row statistics from the LDS-DMA GEMM's per-(token, 64-row block) partials (ONEBIT_FLAG_TILE_STATS): one wave per token,
// parallel-variance combine (Chan et al.); output in onebit_row_stats' format {mean, sum of squared deviations}.
__global__ __launch_bounds__(256) void ob_tile_stats_combine_kernel(const float *__restrict__ tiles, float *__restrict__ stats, int T, int ntiles)
{
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;                                                 // (wave-uniform)
    const ob_float2 *p = reinterpret_cast<const ob_float2 *>(tiles) + (int64_t)t * ntiles;
    float s = 0.f;
    for (int j = lane; j < ntiles; j += 64) s += p[j][0];
    s = ob_wave_sum(s);
    const float mean = s / (64.0f * (float)ntiles);
    float m2 = 0.f;
    for (int j = lane; j < ntiles; j += 64) {
        const ob_float2 v = p[j];
        const float d = v[0] * 0.015625f - mean;
        m2 += v[1] + 64.0f * d * d;
    }
    m2 = ob_wave_sum(m2);
    if (lane == 0) { stats[2 * t] = mean; stats[2 * t + 1] = m2; }
}

template <typename TD>
__global__ __launch_bounds__(256) void ob_normalize_rows_kernel(const TD *__restrict__ u, const float *__restrict__ mean,
                                                                const float *__restrict__ rstd, const TD *__restrict__ bias,
                                                                TD *__restrict__ y, int n)
{
    const int64_t off = (int64_t)blockIdx.x * n;
    const float m = mean[blockIdx.x], r = rstd[blockIdx.x];
    for (int j = threadIdx.x; j < n; j += 256) {
        TD v = (TD)(((float)u[off + j] - m) * r);                   // LayerNorm output in the tensor dtype
        if (bias) v = (TD)((float)v + (float)bias[j]);              // bias added after, in the tensor dtype (:119-120)
        y[off + j] = v;
    }
}
