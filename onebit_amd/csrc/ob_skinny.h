// Skinny 1-bit GEMM for gfx950, 2 <= T <= 64 tokens (short prompts, batched decode):
// z[t][n] = sum_k s[n][k] * fp16(x[t][k] * h[k]) on v_mfma_f32_16x16x32_f16, fp32 accumulate.
//
// At these sizes the packed matrix (N*K/8 bytes from HBM) is the traffic, and what a naive tiling
// drowns in is LDS reads: an activation fragment that feeds one MFMA per read makes the LDS pipe
// (128 B/clk) the bottleneck long before HBM or the matrix pipe.  So:
//   * workgroup = 8 waves = 64 rows x all T tokens (RT groups of 16) x all of K; grid = N / 64;
//   * the 16 k-blocks (32 weights each) of every 512-weight chunk are dealt to the 8 waves (wave w:
//     packed word q = w >> 1, half hf = w & 1, two blocks s2 = 0, 1), and each wave covers ALL FOUR
//     16-row tiles for its blocks: an activation fragment is read from LDS exactly once per
//     workgroup and feeds 4 MFMAs (4 * RT independent accumulators per wave); the 8 partial
//     accumulators meet in LDS at the end;
//   * K advances in phases of PK = 512 elements, double-buffered: the
//     activation tile of a phase (16*RT tokens x PK; every thread owns ONE k-piece of its tokens, so
//     one h load serves them) and the packed rows of the phase (64 rows x PK/8 bytes: 4-8 KB, one
//     coalesced 16-byte load per thread) are loaded one phase ahead into registers, the activations
//     multiplied by h (the fp16 rounding of bitnet.py:113), and both written to padded LDS rows.
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
    float *st;                    // !PARTIAL, optional: per-token LayerNorm partials of u, (sum, M2) per 16-row tile
                                  // (ob_decode.h ObTileStats layout), token stride ob_tile_stats_floats(N); N % 16 == 0
};
__host__ __device__ __forceinline__ int ob_tile_stats_floats(int n) { return ((n + 4095) >> 12) * 512; }
struct ObSkinnyArgs {
    ObSkinnyProj p[3];
    long long ldx;
    int T;
    unsigned long long *dbg;      // -DOB_PROFILE_STAMPS builds: 16 cycle stamps per wave (tools/skinny_phase_probe.py)
};

// tokens x k elements of one phase: 16*RT x 512 (42 / 75 / 142 KB of LDS with the packed rows: three,
// two or one workgroup per CU)
#ifndef OB_SKINNY_PK12
#define OB_SKINNY_PK12 512                      // k elements per phase for 16 / 32-token tiles.  1024 (half the barriers,
                                                // 149 KB of LDS at 32 tokens) measured slower in the batched step: 2.80 vs
                                                // 2.67 ms -- two workgroups no longer share a CU on the wide launches
#endif
#define OB_SKINNY_PK(RT_) ((RT_) >= 4 ? 512 : OB_SKINNY_PK12)
#define OB_SKINNY_PKT(RT_) (OB_SKINNY_PK(RT_) * (RT_))
#define OB_SKINNY_LDS(RT_) OB_SKINNY_LDS2(RT_, 4)
// RNT = 16-row tiles per workgroup: 4 (64 rows), or 8 (128 rows: OB_SKINNY_WIDE=1, measured slower).  (What bounds
// this form is its staging instruction stream between two barriers per phase, not the L2 -- tools/l2_rate_probe.hip;
// ob_skinny3.h is the form without either, for producer-scaled rows.)
#define OB_SKINNY_LDS2(RT_, RNT_) ((size_t)2 * 16 * (RT_) * (OB_SKINNY_PK(RT_) + 8) * 2 + (size_t)2 * 16 * (RNT_) * (OB_SKINNY_PK(RT_) / 32 + 1) * 4)

template <bool PARTIAL, int RT, int RNT = 4>
__global__ __launch_bounds__(512) void ob_skinny_f16_kernel(const ObSkinnyArgs A)
{
    constexpr int ROWS = 16 * RNT;
#ifdef OB_PROFILE_STAMPS
    unsigned long long stamp_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define OB_SK_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp_[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define OB_SK_STAMP(i) do { } while (0)
#endif
    OB_SK_STAMP(0);
    const int pi = (int)blockIdx.x < A.p[0].tile_end ? 0 : ((int)blockIdx.x < A.p[1].tile_end ? 1 : 2);
    const ObSkinnyProj P = pi == 0 ? A.p[0] : (pi == 1 ? A.p[1] : A.p[2]);
    const uint32_t *__restrict__ W = P.W;
    const int64_t ldw_words = P.ldw_words, ldx = A.ldx;
    const _Float16 *__restrict__ x = P.x, *__restrict__ h = P.h, *__restrict__ g = P.g;
    _Float16 *__restrict__ u = P.u;
    float *__restrict__ zp = P.zp;
    float *__restrict__ stp = P.st;
    const int T = A.T, K = P.K, N = P.N;
    const int tile0 = pi == 0 ? 0 : (pi == 1 ? A.p[0].tile_end : A.p[1].tile_end);
    constexpr int PK = OB_SKINNY_PKT(RT) / RT;  // k elements per phase
    constexpr int CPP = PK / 512;               // 512-weight chunks per phase
    constexpr int TT = 16 * RT;                 // tokens of the tile
    constexpr int PITCH = PK + 8;               // halves per activation row (16-byte pad: rows shift by 4 banks)
    constexpr int WP = PK / 32 + 1;             // dwords per packed row in LDS (+1: the 16 rows of a tile hit 16 banks)
    constexpr int PPR = PK / 8;                 // 16-byte activation pieces per token row and phase
    constexpr int WPR = PK / 128;               // 16-byte packed pieces per weight row and phase
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *As = reinterpret_cast<_Float16 *>(smem);                                   // [2][TT][PITCH]
    uint32_t *Ws = reinterpret_cast<uint32_t *>(smem + (size_t)2 * TT * PITCH * 2);      // [2][ROWS][WP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, gq = lane >> 4;
    const int wq = wave >> 1, hf = wave & 1;    // this wave's packed word and half of every chunk
    const int n0 = ((int)blockIdx.x - tile0) * ROWS;
    const int nph = (K + PK - 1) / PK;
    const int nwords = K >> 5;

    // staging: thread owns activation k-piece `kp` of token rows tok0 + TSTEP * i, and (threads below
    // 64 * WPR) one 16-byte piece of one packed row
    const int kp = tid % PPR, tok0 = tid / PPR;
    constexpr int TSTEP = 512 / PPR;            // 8 token rows per pass of the 512 threads
    constexpr int NSTG = TT / TSTEP;            // 2, 4, 8 loads per thread and phase for RT = 1, 2, 4
    const _Float16 *xrow[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; ++i) xrow[i] = x + (int64_t)min(tok0 + TSTEP * i, T - 1) * ldx;
    const bool wload = tid < ROWS * WPR;
    const int wrow_i = wload ? tid / WPR : 0, wpc = wload ? tid % WPR : 0;
    const uint32_t *wsrc = W + (int64_t)min(n0 + wrow_i, N - 1) * ldw_words;

    ob_float4 acc[RNT][RT];
#pragma unroll
    for (int a = 0; a < RNT; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) acc[a][b] = (ob_float4){0.f, 0.f, 0.f, 0.f};

    // raw loads only (clamped addresses); masks are applied when the registers are consumed
    ob_half8 xs[NSTG], hs;
    ob_u32x4 wv = {0u, 0u, 0u, 0u};
    bool kv_ld = true, wv_ld = true;
    auto load_phase = [&](int ph) {
        const int k = ph * PK + kp * 8;
        kv_ld = k < K;
        const int kc = kv_ld ? k : 0;
        hs = *reinterpret_cast<const ob_half8 *>(h + kc);
#pragma unroll
        for (int i = 0; i < NSTG; ++i) xs[i] = *reinterpret_cast<const ob_half8 *>(xrow[i] + kc);
        const int word = ph * (PK / 32) + wpc * 4;
        wv_ld = word < nwords;
        if (wload) wv = __builtin_nontemporal_load(reinterpret_cast<const ob_u32x4 *>(wsrc + min(word, nwords - 4)));
    };
    auto store_phase = [&](int buf) {
        _Float16 *dst = As + (size_t)buf * TT * PITCH + kp * 8;
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            ob_half8 a = xs[i] * hs;                                                   // fp16(x*h)
            if (!kv_ld) a = (ob_half8)(_Float16)0;
            *reinterpret_cast<ob_half8 *>(dst + (size_t)(tok0 + TSTEP * i) * PITCH) = a;
        }
        if (wload) {
            uint32_t *wd = Ws + ((size_t)buf * ROWS + wrow_i) * WP + wpc * 4;
            const ob_u32x4 w = wv_ld ? wv : (ob_u32x4){0u, 0u, 0u, 0u};
            wd[0] = w[0]; wd[1] = w[1]; wd[2] = w[2]; wd[3] = w[3];                    // WP is odd: dword stores
        }
    };

    // The loads of phase ph + 1 are issued BEFORE the barrier that ends phase ph - 1 (right behind the LDS stores
    // that emptied the staging registers): a workgroup is alone on its CU for most layer shapes, and issued at the
    // top of phase ph they had only its ~1500 cycles of MFMA work to cover ~1900 cycles of latency -- every phase
    // then ended parked on them (tools/skinny_phase_probe.py: 2500 cycles per phase).
    load_phase(0);
    OB_SK_STAMP(1);
    store_phase(0);
    if (nph > 1) load_phase(1);
    __syncthreads();
    OB_SK_STAMP(2);

    for (int ph = 0; ph < nph; ++ph) {
        const int cur = ph & 1;
        const bool more = ph + 1 < nph;
        __builtin_amdgcn_sched_barrier(0);
        const _Float16 *Ab = As + (size_t)cur * TT * PITCH + (size_t)r * PITCH + gq * 128 + wq * 32 + hf * 16;
        const uint32_t *Wb = Ws + ((size_t)cur * ROWS + r) * WP + gq * 4 + wq;
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            uint32_t e[RNT][8];
#pragma unroll
            for (int rn = 0; rn < RNT; ++rn)
                ob_expand16((Wb[(size_t)rn * 16 * WP + c * 16] >> (16 * hf)) & 0xffffu, e[rn]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                ob_half8 bop[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    bop[rt] = *reinterpret_cast<const ob_half8 *>(Ab + (size_t)rt * 16 * PITCH + c * 512 + s2 * 8);
#pragma unroll
                for (int rn = 0; rn < RNT; ++rn) {
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
        if (ph == 0) OB_SK_STAMP(3);                        // MFMAs of phase 0 issued
        if (more) store_phase(cur ^ 1);
        if (ph + 2 < nph) load_phase(ph + 2);
        __syncthreads();
        if (ph == 0) OB_SK_STAMP(4);
        if (ph == 1) OB_SK_STAMP(5);
        if (ph == 3) OB_SK_STAMP(6);
    }
    OB_SK_STAMP(7);

    // the 8 waves' partial accumulators meet in LDS (the staging buffers are free after the last barrier),
    // four row tiles per pass: [wave][rn][rt][lane] float4, then output slot (rn, rt, lane) is summed by one thread
    ob_float4 *zr = reinterpret_cast<ob_float4 *>(smem);
    constexpr int NSLOT = 4 * RT * 64;
#pragma unroll
    for (int pass = 0; pass < RNT / 4; ++pass) {
        if (pass) __syncthreads();                          // the previous pass's sums have been read
#pragma unroll
        for (int rn = 0; rn < 4; ++rn)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) zr[((wave * 4 + rn) * RT + rt) * 64 + lane] = acc[4 * pass + rn][rt];
        __syncthreads();
        if (pass == 0) OB_SK_STAMP(8);
        for (int slot = tid; slot < NSLOT; slot += 512) {
            const int sl = slot & 63, rt = (slot >> 6) % RT, rn = (slot >> 6) / RT;
            ob_float4 z = zr[((0 * 4 + rn) * RT + rt) * 64 + sl];
#pragma unroll
            for (int w = 1; w < 8; ++w) z += zr[((w * 4 + rn) * RT + rt) * 64 + sl];
            const int t = rt * 16 + (sl & 15);
            const int ntile = n0 + (4 * pass + rn) * 16;
            const int nb = ntile + 4 * (sl >> 4);
            if (!PARTIAL && stp) {
                // the consumer's LayerNorm partials: the 16 rows of the tile for token t live in the lanes
                // sl, sl ^ 16, sl ^ 32, sl ^ 48 (4 rows each); every lane of the wave is here (uniform trip count)
                float o4[4], sm = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o4[i] = (float)(_Float16)(ob_round_h(z[i]) * (float)g[min(nb + i, N - 1)]);
                    sm += o4[i];
                }
                sm += __shfl_xor(sm, 16);
                sm += __shfl_xor(sm, 32);
                const float mu = sm * 0.0625f;
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) m2 = __builtin_fmaf(o4[i] - mu, o4[i] - mu, m2);
                m2 += __shfl_xor(m2, 16);
                m2 += __shfl_xor(m2, 32);
                if (sl < 16 && t < T && ntile < N) {
                    float *d = stp + (size_t)t * ob_tile_stats_floats(N) + (size_t)(ntile >> 4) * 2;
                    d[0] = sm; d[1] = m2;
                }
            }
            if (t >= T) continue;
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
#ifdef OB_PROFILE_STAMPS
    OB_SK_STAMP(9);
    if (A.dbg && (threadIdx.x & 63) == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i_ = 0; i_ < 16; ++i_) A.dbg[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + i_] = stamp_[i_];
    }
#endif
#undef OB_SK_STAMP
}
