// Batched-prefill 1-bit GEMM for gfx950 (T >= 17 tokens): z[t][n] = sum_k s[n][k] * fp16(x[t][k]*h[k])
// on v_mfma_f32_16x16x32_f16, fp32 accumulate (the arithmetic of bitnet.py:113-115).
//
// Prefill workgroup tile 128 rows (n) x 128 tokens (t), 4 waves as 2 (n) x 2 (t), each wave 64 x 64 =
// 4 x 4 MFMA tiles (64 accumulator VGPRs); smaller token tiles for 2 <= T <= 64 (template below).
// K advances 128 per step:
//   * activations: global -> registers -> (x * h, one v_pk_mul_f16 per pair = the fp16 rounding of
//     bitnet.py:113) -> LDS, double buffered, rows padded to 272 B so the 16 lanes of a
//     ds_read_b128 group (16 different tokens, same k) fall on distinct banks;
//   * weights: never through LDS -- each lane loads ONE packed dword per 16-row tile per step
//     (its row, 32 of the 128 k) straight into registers and expands it to +-1.0 fp16 MFMA
//     A operands; one expansion feeds 4 token tiles, so the VALU cost is amortised 4x.
// The packed matrix is 1/16 of the activation bytes here; the kernel is MFMA-bound, not HBM-bound.
// Workgroups are renumbered so that consecutive ids on one XCD share the token tile (its 1 MB of
// activations stays in that XCD's L2).
#pragma once
#include "ob_common.h"

#define OB_GB_N 128
#define OB_GB_T 128
#define OB_GB_K 128
#define OB_GB_PITCH 136      // halves per LDS row: 128 + 8 (272 B)

// Tile shape: WN x WT waves, each wave RN x RT MFMA tiles of 16 rows x 16 tokens.
//   <2,2,4,4>: 128 rows x 128 tokens, the prefill GEMM (2 workgroups per CU).
//   <4,1,1,RT>: 64 rows x 16*RT tokens for 2 <= T <= 64 ("skinny": short prompts, batched decode):
//               no MFMA work on token padding, the activation tile of a step is 4-16 KB, and the grid has
//               N/64 workgroups so the packed matrix streams from many CUs.
template <bool PARTIAL, int WN, int WT, int RN, int RT>
__global__ __launch_bounds__(WN * WT * 64, (WN * WT * RN * RT >= 64) ? 2 : 1) void ob_gemm_f16_kernel(
    const uint32_t *__restrict__ W, int64_t ldw_words, const _Float16 *__restrict__ x, int64_t ldx,
    const _Float16 *__restrict__ h, const _Float16 *__restrict__ g, _Float16 *__restrict__ u,
    float *__restrict__ zp, int T, int K, int N, int nbn, int nbt)
{
    constexpr int NTH = WN * WT * 64, TILE_N = WN * RN * 16, TILE_T = WT * RT * 16;
    constexpr int NST = TILE_T * 16 / NTH;                  // 8-half staging chunks per thread and K step
    static_assert(NST >= 1 && NST * NTH == TILE_T * 16, "staging does not tile");
    __shared__ __attribute__((aligned(16))) _Float16 As[2][TILE_T][OB_GB_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wt = wave / WN;
    const int r = lane & 15, gq = lane >> 4;

    // XCD-aware renumbering (bijective for any grid size): XCD x owns a contiguous id range
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
    const int bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    const int tt = bid / nbn, tn = bid - tt * nbn;          // n fastest: neighbours share the token tile
    const int n0 = tn * TILE_N, t0 = tt * TILE_T;
    (void)nbt;

    const int nwords = K >> 5;
    const int nk = (K + OB_GB_K - 1) / OB_GB_K;

    // staging: thread -> (token st_t + 16 i, halves st_k .. st_k + 7), 16 lanes cover 256 contiguous bytes
    const int st_t = tid >> 4, st_k = (tid & 15) * 8;
    const _Float16 *xrow[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) xrow[i] = x + (int64_t)min(t0 + st_t + (NTH / 16) * i, T - 1) * ldx;

    // weights: row of tile rn for this lane
    const uint32_t *wrow[RN];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) wrow[rn] = W + (int64_t)min(n0 + (wn * RN + rn) * 16 + r, N - 1) * ldw_words;

    ob_float4 acc[RN][RT];
#pragma unroll
    for (int a = 0; a < RN; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) acc[a][b] = (ob_float4){0.f, 0.f, 0.f, 0.f};

    ob_half8 xs[NST], hs;
    uint32_t wcur[RN], wnext[RN];

    // Loads are raw (addresses clamped into the arrays); nothing touches a loaded register until the
    // MFMA block of the current step has been issued -- a select right behind a load would park the
    // wave on vmcnt(0) (loads return in order) for a full L2 round trip in every K step.
    bool kv_ld = true, wv_ld = true;
    auto load_step = [&](int ks) {
        const int k = ks * OB_GB_K + st_k;
        kv_ld = k < K;                                        // K % 8 == 0: the 8 halves are all in or all out
        const int kc = kv_ld ? k : 0;
        hs = *reinterpret_cast<const ob_half8 *>(h + kc);
#pragma unroll
        for (int i = 0; i < NST; ++i) xs[i] = *reinterpret_cast<const ob_half8 *>(xrow[i] + kc);
    };
    auto load_w = [&](int ks, uint32_t (&w)[RN]) {
        const int word = ks * 4 + gq;
        wv_ld = word < nwords;
        const int wc = min(word, nwords - 1);
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) w[rn] = wrow[rn][wc];
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            ob_half8 a = xs[i] * hs;                                                   // fp16(x*h)
            if (!kv_ld) a = (ob_half8)(_Float16)0;
            *reinterpret_cast<ob_half8 *>(&As[buf][st_t + (NTH / 16) * i][st_k]) = a;
        }
    };

    load_step(0);
    load_w(0, wcur);
    store_step(0);
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) wcur[rn] = wv_ld ? wcur[rn] : 0u;
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nk;
#if defined(OB_GEMM_ABL) && (OB_GEMM_ABL & 4)
        if (more && ks == 0) {                                                         // no global traffic after step 1
#else
        if (more) {
#endif
            load_step(ks + 1);
            load_w(ks + 1, wnext);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            uint32_t e[RN][8];
#pragma unroll
            for (int rn = 0; rn < RN; ++rn) {
#if defined(OB_GEMM_ABL) && (OB_GEMM_ABL & 1)
#pragma unroll
                for (int i = 0; i < 8; ++i) e[rn][i] = wcur[rn] + i + hf;          // no sign expansion
#else
                ob_expand16((wcur[rn] >> (16 * hf)) & 0xffffu, e[rn]);
#endif
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int s = 2 * hf + s2;
                ob_half8 bop[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#if defined(OB_GEMM_ABL) && (OB_GEMM_ABL & 2)
                    bop[rt] = xs[rt % NST];                                            // no LDS operand reads
#else
                    bop[rt] = *reinterpret_cast<const ob_half8 *>(&As[cur][(wt * RT + rt) * 16 + r][gq * 32 + 8 * s]);
#endif
                }
#pragma unroll
                for (int rn = 0; rn < RN; ++rn) {
                    ob_u32x4 av = {e[rn][4 * s2 + 0], e[rn][4 * s2 + 1], e[rn][4 * s2 + 2], e[rn][4 * s2 + 3]};
                    ob_half8 aop;
                    __builtin_memcpy(&aop, &av, 16);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[rn][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bop[rt], acc[rn][rt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            store_step(cur ^ 1);
#pragma unroll
            for (int rn = 0; rn < RN; ++rn) wcur[rn] = wv_ld ? wnext[rn] : 0u;
        }
        __syncthreads();
    }

    // epilogue: D[n][t]: lane holds n = 4*gq + i, t = r of each 16 x 16 tile
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int t = t0 + (wt * RT + rt) * 16 + r;
        if (t >= T) continue;
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
            const int nb = n0 + (wn * RN + rn) * 16 + 4 * gq;
            if (PARTIAL) {
                if (nb + 3 < N && (N & 3) == 0) {
                    *reinterpret_cast<ob_float4 *>(zp + (int64_t)t * N + nb) = acc[rn][rt];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (nb + i < N) zp[(int64_t)t * N + nb + i] = acc[rn][rt][i];
                }
            } else {
                _Float16 o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gn = (float)g[min(nb + i, N - 1)];
                    o[i] = (_Float16)(ob_round_h(acc[rn][rt][i]) * gn);      // fp16(z) (:115), * g -> fp16 (:116)
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
}
