// Skinny 1-bit GEMM for gfx950, 2 <= T <= 64 tokens (short prompts, batched decode):
// z[t][n] = sum_k s[n][k] * fp16(x[t][k] * h[k]) on v_mfma_f32_16x16x32_f16, fp32 accumulate.
//
// At these sizes the packed matrix (N*K/8 bytes from HBM) is the traffic and latency is the enemy,
// so the kernel is built from few, large, fully prefetched phases instead of many small K steps:
//   * workgroup = 8 waves = 64 rows x all T tokens (RT groups of 16) x all of K: wave (wr, kh) owns
//     the 16-row tile wr and the k-half kh of every 512-weight chunk (words 2kh, 2kh+1 of the lane's
//     four), so two waves per SIMD overlap sign expansion, MFMA and LDS reads; the two halves meet
//     in LDS at the end.  grid = N / 64 workgroups, one per CU;
//   * K advances in phases of PK = 1024 / RT (RT = 4: 512) elements: the activation
//     tile of a phase (16*RT tokens x PK) is loaded coalesced into registers one phase ahead
//     (4-8 x 16-byte loads per thread, every thread owns ONE k-piece of its tokens, so one h load
//     serves them), multiplied by h (the fp16 rounding of bitnet.py:113) and written to padded LDS
//     rows; the weights of a phase are PK/512 dwordx4 per lane, also one phase ahead;
//   * per 512-weight chunk a wave expands its 16 rows' signs once and feeds RT token groups;
//     with fewer than 4 token groups the k-blocks rotate over 4 / RT accumulators per group so
//     that consecutive MFMAs never wait on each other.
// Same register-level conventions as ob_gemm.h / ob_decode.h: weights = A operand (row = lane & 15,
// k-group = lane >> 4), activations = B operand (column = token).
#pragma once
#include "ob_common.h"

// Up to 3 projections that share the activations (q|k|v, gate|up) ride in one launch: workgroup b
// belongs to the projection whose tile range contains b.
// An entry may also be a K-slice of a projection (W, h, x advanced to the slice, K = its length,
// PARTIAL output): split-K over workgroups for short-and-wide layers (down_proj), summed by the consumer.
struct ObSkinnyProj {
    const uint32_t *W; long long ldw_words;
    const _Float16 *h, *g;
    const _Float16 *x;            // activations [T, ldx] (already advanced to the K-slice)
    _Float16 *u;                  // !PARTIAL: fp16 [T, N]
    float *zp;                    // PARTIAL: fp32 sums [T, N]
    int N, K, tile_end;           // tiles [previous tile_end, tile_end) of the grid
};
struct ObSkinnyArgs {
    ObSkinnyProj p[3];
    long long ldx;
    int T;
};

// tokens x k elements of one phase: 16 x 1024 / 32 x 512 (66 KB of LDS: two workgroups per CU, so a
// grid of up to 512 tiles is resident at once and one workgroup's loads hide behind the other's
// MFMAs) or 64 x 512 (133 KB, one workgroup per CU)
#define OB_SKINNY_PKT(RT_) ((RT_) == 4 ? 2048 : 1024)

template <bool PARTIAL, int RT>
__global__ __launch_bounds__(512) void ob_skinny_f16_kernel(const ObSkinnyArgs A)
{
    const int pi = (int)blockIdx.x < A.p[0].tile_end ? 0 : ((int)blockIdx.x < A.p[1].tile_end ? 1 : 2);
    const ObSkinnyProj P = pi == 0 ? A.p[0] : (pi == 1 ? A.p[1] : A.p[2]);
    const uint32_t *__restrict__ W = P.W;
    const int64_t ldw_words = P.ldw_words, ldx = A.ldx;
    const _Float16 *__restrict__ x = P.x, *__restrict__ h = P.h, *__restrict__ g = P.g;
    _Float16 *__restrict__ u = P.u;
    float *__restrict__ zp = P.zp;
    const int T = A.T, K = P.K, N = P.N;
    const int tile0 = pi == 0 ? 0 : (pi == 1 ? A.p[0].tile_end : A.p[1].tile_end);
    constexpr int PK = OB_SKINNY_PKT(RT) / RT;  // k elements per phase
    constexpr int CPP = PK / 512;               // 512-weight chunks (one dwordx4 per lane) per phase
    constexpr int TT = 16 * RT;                 // tokens of the tile
    constexpr int PITCH = PK + 8;               // halves per LDS row (16-byte pad: rows shift by 4 banks)
    constexpr int NS = 4 / RT;                  // accumulators per token group
    constexpr int PPR = PK / 8;                 // 16-byte pieces per token row and phase
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *As = reinterpret_cast<_Float16 *>(smem);          // [2][TT][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, gq = lane >> 4;
    const int wr = wave & 3, kh = wave >> 2;
    const int n0 = ((int)blockIdx.x - tile0) * 64 + wr * 16;
    const int nph = (K + PK - 1) / PK;
    const int nwords = K >> 5;

    // staging: thread owns k-piece `kp` of token rows tok0 + (512 / PPR) * i
    const int kp = tid % PPR, tok0 = tid / PPR;
    constexpr int TSTEP = 512 / PPR;            // 2, 4, 8 for RT = 1, 2, 4
    constexpr int NSTG = TT / TSTEP;            // 4 (RT = 1, 2) or 8 (RT = 4) loads per thread and phase
    const _Float16 *xrow[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; ++i) xrow[i] = x + (int64_t)min(tok0 + TSTEP * i, T - 1) * ldx;
    const uint32_t *wrow = W + (int64_t)min(n0 + r, N - 1) * ldw_words;

    ob_float4 acc[RT][NS];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < NS; ++b) acc[a][b] = (ob_float4){0.f, 0.f, 0.f, 0.f};

    // raw loads only (clamped addresses); masks are applied after the phase's MFMA block is issued
    ob_half8 xs[NSTG], hs;
    ob_u32x2 wcur[CPP], wnext[CPP];
    bool kv_ld = true;
    auto load_x = [&](int ph) {
        const int k = ph * PK + kp * 8;
        kv_ld = k < K;
        const int kc = kv_ld ? k : 0;
        hs = *reinterpret_cast<const ob_half8 *>(h + kc);
#pragma unroll
        for (int i = 0; i < NSTG; ++i) xs[i] = *reinterpret_cast<const ob_half8 *>(xrow[i] + kc);
    };
    auto load_w = [&](int ph, ob_u32x2 (&w)[CPP]) {
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            const int word = (ph * CPP + c) * 16 + gq * 4;
            w[c] = __builtin_nontemporal_load(reinterpret_cast<const ob_u32x2 *>(wrow + min(word, nwords - 4) + 2 * kh));
        }
    };
    auto mask_w = [&](int ph, ob_u32x2 (&w)[CPP]) {
#pragma unroll
        for (int c = 0; c < CPP; ++c)
            if ((ph * CPP + c) * 16 + gq * 4 >= nwords) w[c] = (ob_u32x2){0u, 0u};
    };
    auto store_x = [&](int buf) {
        _Float16 *dst = As + (size_t)buf * TT * PITCH + kp * 8;
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            ob_half8 a = xs[i] * hs;                                                   // fp16(x*h)
            if (!kv_ld) a = (ob_half8)(_Float16)0;
            *reinterpret_cast<ob_half8 *>(dst + (size_t)(tok0 + TSTEP * i) * PITCH) = a;
        }
    };

    load_x(0);
    load_w(0, wcur);
    store_x(0);
    mask_w(0, wcur);
    __syncthreads();

    for (int ph = 0; ph < nph; ++ph) {
        const int cur = ph & 1;
        const bool more = ph + 1 < nph;
        if (more) {
            load_x(ph + 1);
            load_w(ph + 1, wnext);
        }
        __builtin_amdgcn_sched_barrier(0);
        const _Float16 *Ab = As + (size_t)cur * TT * PITCH + (size_t)r * PITCH + gq * 128;
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int q = 2 * kh + q2;          // this wave's words of the chunk
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    uint32_t e[8];
                    ob_expand16((wcur[c][q2] >> (16 * hf)) & 0xffffu, e);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        ob_u32x4 av = {e[4 * s2 + 0], e[4 * s2 + 1], e[4 * s2 + 2], e[4 * s2 + 3]};
                        ob_half8 aop;
                        __builtin_memcpy(&aop, &av, 16);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const ob_half8 bop = *reinterpret_cast<const ob_half8 *>(
                                Ab + (size_t)rt * 16 * PITCH + c * 512 + q * 32 + (2 * hf + s2) * 8);
                            acc[rt][(2 * hf + s2) % NS] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bop, acc[rt][(2 * hf + s2) % NS], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            store_x(cur ^ 1);
            mask_w(ph + 1, wnext);
#pragma unroll
            for (int c = 0; c < CPP; ++c) wcur[c] = wnext[c];
        }
        __syncthreads();
    }

    // the two k-halves meet in LDS (the activation buffers are free after the last barrier)
    float *zr = reinterpret_cast<float *>(smem);                // [4 row tiles][RT][64 lanes][4]
    if (kh == 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            ob_float4 z = acc[rt][0];
#pragma unroll
            for (int b = 1; b < NS; ++b) z += acc[rt][b];
            *reinterpret_cast<ob_float4 *>(zr + ((size_t)(wr * RT + rt) * 64 + lane) * 4) = z;
        }
    }
    __syncthreads();
    if (kh == 1) return;
    // epilogue: lane holds rows n0 + 4*gq + i of token 16*rt + r
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        ob_float4 z = acc[rt][0];
#pragma unroll
        for (int b = 1; b < NS; ++b) z += acc[rt][b];
        z += *reinterpret_cast<const ob_float4 *>(zr + ((size_t)(wr * RT + rt) * 64 + lane) * 4);
        const int t = rt * 16 + r;
        if (t >= T) continue;
        const int nb = n0 + 4 * gq;
        if (PARTIAL) {
            if (nb + 3 < N && (N & 3) == 0) {
                *reinterpret_cast<ob_float4 *>(zp + (int64_t)t * N + nb) = z;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (nb + i < N) zp[(int64_t)t * N + nb + i] = z[i];
            }
        } else {
            _Float16 o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float gn = (float)g[min(nb + i, N - 1)];
                o[i] = (_Float16)(ob_round_h(z[i]) * gn);                  // fp16(z) (:115), * g -> fp16 (:116)
            }
            if (nb + 3 < N && (N & 3) == 0) {
                ob_half4 ov = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<ob_half4 *>(u + (int64_t)t * N + nb) = ov;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (nb + i < N) u[(int64_t)t * N + nb + i] = o[i];
            }
        }
    }
}
