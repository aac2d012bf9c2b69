// LDS-DMA prefill GEMM on v_mfma_f32_32x32x16_f16 (round 6).  The tiling, the staging and the pipeline of ob_gemm3_f16_kernel<*, 1>
// (ob_gemm2.h: 256 rows x 128 tokens per 4-wave workgroup, two workgroups per CU, pre-scaled activation tiles and packed rows global ->
// LDS by DMA, four tile buffers, ONE raw barrier per K step behind a counted vmcnt) with the matrix instruction of twice the shape:
//   * tools/mfma32_probe.hip: 16x16x32 and 32x32x16 sustain the same flops per cycle bare (17.1 / 16.4 cycles per 16x16x32-equivalent), but
//     with the 2-4 VALU operations per equivalent that the sign expansion costs, the 16x16x32 stream drops to 35-37 cycles per
//     equivalent while the 32x32x16 stream stays at 17-21: an MFMA occupies the SIMD's issue port for a fixed ~8 cycles whatever its
//     size, so half as many instructions for the same flops leave twice the room for the expansion's VALU work.  The 16x16 kernel sat
//     at 70 % MFMA busy with 2.5 VALU per MFMA (profiles/r05_pmc_prefill_mfma.txt).
//   * operand layout (verified by the probe): A = weights, lane l: row l % 32, k = 8 (l / 32) + j; B = activations, lane l: token l % 32,
//     the same k; D[i]: row 8 (i / 4) + 4 (l / 32) + i % 4, token l % 32 -- four consecutive rows of one token per accumulator quad, one
//     8-byte store into row-major y[T, N] as before.
//   * a K step (64 k) is four sub-steps of 16 k; sub-step s, lane half kg = l / 32 uses LOGICAL chunk c = 2 s + kg of the tile row (the
//     k order of a step is the 16x16 kernel's: chunk c of step 2 m + par covers k = 128 m + 64 (c >> 2) + 32 par + 8 (c & 3) ..+ 7), so
//     its 8 sign bits are byte 2 (s & 1) + kg of word 2 (s >> 1) + par of the row's 16 bytes per PAIR of steps: one ds_read_b128 per
//     32-row tile and pair (lanes l and l + 32 read the same address: broadcast), one v_bfe + 3 + 8 VALU per operand of 4 MFMAs.
//   * LDS swizzle of the activation tile: chunk c of row t at c ^ f(t), f(t) = bit 1 | bit 3 << 1 | bit 4 << 2 of t: the 16 lanes of
//     every ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32) are 8 even and 8 odd rows reading ONE
//     logical chunk; f is a bijection onto 0..7 on each of those row sets, so a group covers the 64 banks exactly once.
// Wave tile 64 rows x 128 tokens = 2 x 4 tiles of 32 x 32 (128 accumulator registers, as before).
#pragma once
#include "ob_gemm2.h"

typedef float ob_float16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int ob_g4_swz(int t, int c) { return c ^ (((t >> 1) & 1) | (((t >> 3) & 1) << 1) | (((t >> 4) & 1) << 2)); }

#define OB_G4_LDS (OB_G3_BUFS * 128 * OB_G2_PITCH * 2 + 2 * 8192)      // 4 activation tiles (16 KB each) + 2 weight quads

template <bool PARTIAL>
__device__ __forceinline__ void ob_gemm4_body(
    const uint32_t *__restrict__ W, int64_t ldw_words, const _Float16 *__restrict__ a, int64_t lda,
    const _Float16 *__restrict__ g, _Float16 *__restrict__ u, float *__restrict__ zp, int T, int K, int N, int nbn, const int bid)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TTILE = 128;
    _Float16 (*As)[TTILE][OB_G2_PITCH] = reinterpret_cast<_Float16 (*)[TTILE][OB_G2_PITCH]>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave;                          // 4 waves: 64 rows each, all 128 tokens
    const int r32 = lane & 31, kg = lane >> 5;
    const int tt = bid / nbn, tn = bid - tt * nbn;
    const int n0 = tn * OB_G2_N, t0 = tt * TTILE;
    const int nk = K / OB_G2_K;                   // K % 256 == 0 (host-checked): whole quads of steps

    uint32_t src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = ob_g4_swz(row, lane & 7);
        src[i] = (uint32_t)(((int64_t)min(t0 + row, T - 1) * lda + (c >> 2) * 64 + (c & 3) * 8) * 2);
    }
    const char *abase = reinterpret_cast<const char *>(a);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    auto dma16 = [&](const char *base, uint32_t voff, uint32_t lds_addr) {
        uint32_t m0_keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_keep) : "s"(lds_addr), "v"(voff), "s"(base) : "memory");
    };
    auto dma = [&](int tile) {
        const uint32_t buf = (uint32_t)tile & (OB_G3_BUFS - 1);
        const uint32_t kofs = (uint32_t)((tile >> 1) * 128 + (tile & 1) * 32) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dma16(abase, src[i] + kofs, lds0 + (buf * TTILE + (wave_u * 4 + i) * 8) * (OB_G2_PITCH * 2));
    };
    const char *wbase = reinterpret_cast<const char *>(W);
    const uint32_t wsrc = (uint32_t)((int64_t)min(n0 + 32 * wave + (lane & 31), N - 1) * ldw_words * 4 + 16 * (lane >> 5));
    const uint32_t wsrc2 = (uint32_t)((int64_t)min(n0 + 32 * (wave + 4) + (lane & 31), N - 1) * ldw_words * 4 + 16 * (lane >> 5));
    constexpr uint32_t WOFF = OB_G3_BUFS * TTILE * OB_G2_PITCH * 2;
    char *wlds = smem + WOFF;
    auto wdma = [&](int quad) {
        dma16(wbase, wsrc + 32u * (uint32_t)quad, lds0 + WOFF + ((uint32_t)quad & 1) * 8192 + wave_u * 1024);
        dma16(wbase, wsrc2 + 32u * (uint32_t)quad, lds0 + WOFF + ((uint32_t)quad & 1) * 8192 + (wave_u + 4) * 1024);
    };
    // this lane's 16 bytes (a pair of steps) of row r32 of the wave's two 32-row blocks 2 wn, 2 wn + 1
    const char *wrd = wlds + wn * 2048 + r32 * 16;
    const uint32_t boff0 = 8u * (uint32_t)kg, boff1 = boff0 + 16u;        // bit offset of the lane's byte inside a word: sub-step even / odd

    ob_float16 acc[2][4];
#pragma unroll
    for (int x_ = 0; x_ < 2; ++x_)
#pragma unroll
        for (int y_ = 0; y_ < 4; ++y_)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[x_][y_][i] = 0.f;
    wdma(0);
    dma(0);
    dma(1);
    dma(2);
    dma(3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ob_u32x4 wc[2], wnx[2];
#pragma unroll
    for (int rn = 0; rn < 2; ++rn) {
        wc[rn] = *reinterpret_cast<const ob_u32x4 *>(wrd + rn * 1024);
        wnx[rn] = wc[rn];
    }
    ob_half8 bopA[2][4], bopB[2][4];             // operands of sub-steps {0, 1} / {2, 3}: the whole step's operands are in registers
#define OB_G4_READ(DST, BUF, SUB)                                                                                        \
    _Pragma("unroll") for (int t4 = 0; t4 < 4; ++t4)                                                                     \
        DST[t4] = *reinterpret_cast<const ob_half8 *>(&As[BUF][t4 * 32 + r32][ob_g4_swz(r32, 2 * (SUB) + kg) * 8]);
    // 8 sign bits -> 4 dwords of packed +-1.0 (ob_expand16's arithmetic on one byte), then 4 MFMAs per row tile
#define OB_G4_MMA(SRC, SUB, PAR)                                                                                         \
    _Pragma("unroll") for (int rn = 0; rn < 2; ++rn) {                                                                   \
        const uint32_t b8 = __builtin_amdgcn_ubfe(wc[rn][2 * ((SUB) >> 1) + (PAR)], ((SUB) & 1) ? boff1 : boff0, 8u);   \
        const uint32_t cw = (b8 & 0x55u) | ((b8 & 0xAAu) << 15);                                                         \
        ob_u32x4 av;                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) av[p] = ((cw << (15 - 2 * p)) & mask) | (0x3C003C00u & ~mask);     \
        const ob_half8 aop = __builtin_bit_cast(ob_half8, av);                                                           \
        _Pragma("unroll") for (int t4 = 0; t4 < 4; ++t4)                                                                 \
            acc[rn][t4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, SRC[t4], acc[rn][t4], 0, 0, 0);                    \
    }
    int ks = -1;
    OB_G4_READ(bopA[0], 0, 0)
    OB_G4_READ(bopA[1], 0, 1)

    // One K step (Q = step within its quad, STEADY: literals), the schedule of ob_gemm3's step with half-steps of two sub-steps:
    //   read the operands of sub-steps 2, 3 | 16 MFMAs of sub-steps 0, 1 | COUNTED wait + raw barrier (tile kc + 1 has landed for
    //   everyone -- transfers complete in order: at most the transfers of steps kc - 2 and kc - 1 are outstanding, 4 + 4 + the two
    //   weight DMAs if either opens a quad --, and every read of tile kc is complete) | [Q == 0: weights of the next quad] DMA of tile
    //   kc + 4 into tile kc's buffer | [Q odd: the next pair's packed words] operands of sub-steps 0, 1 of step kc + 1 | 16 MFMAs of
    //   sub-steps 2, 3.
#define OB_G4_STEP(Q, STEADY)                                                                                            \
    {                                                                                                                    \
        const int kc = ks + (Q), cur = kc & (OB_G3_BUFS - 1);                                                            \
        uint32_t mask = 0x80008000u;                                                                                     \
        asm("" : "+s"(mask));                                                                                            \
        OB_G4_READ(bopB[0], cur, 2)                                                                                      \
        OB_G4_READ(bopB[1], cur, 3)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        OB_G4_MMA(bopA[0], 0, (Q) & 1)                                                                                   \
        OB_G4_MMA(bopA[1], 1, (Q) & 1)                                                                                   \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (STEADY) {                                                                                                    \
            if ((Q) == 1 || (Q) == 2) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");          \
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                \
        } else {                                                                                                         \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                     \
        }                                                                                                                \
        if (STEADY) {                                                                                                    \
            if ((Q) == 0) wdma((kc >> 2) + 1);                                                                           \
            dma(kc + 4);                                                                                                 \
        }                                                                                                                \
        if (((Q) & 1) && ((STEADY) || kc + 1 < nk)) {                                                                    \
            const int pm = (kc + 1) >> 1;                                                                                \
            const char *wq = wrd + ((pm >> 1) & 1) * 8192 + (pm & 1) * 512;                                              \
            _Pragma("unroll") for (int rn = 0; rn < 2; ++rn) wnx[rn] = *reinterpret_cast<const ob_u32x4 *>(wq + rn * 1024);  \
        }                                                                                                                \
        if ((STEADY) || kc + 1 < nk) {                                                                                   \
            OB_G4_READ(bopA[0], (kc + 1) & (OB_G3_BUFS - 1), 0)                                                          \
            OB_G4_READ(bopA[1], (kc + 1) & (OB_G3_BUFS - 1), 1)                                                          \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        OB_G4_MMA(bopB[0], 2, (Q) & 1)                                                                                   \
        OB_G4_MMA(bopB[1], 3, (Q) & 1)                                                                                   \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if ((Q) & 1) {                                                                                                   \
            asm volatile("" : "+v"(wnx[0]), "+v"(wnx[1]));                                                               \
            _Pragma("unroll") for (int rn = 0; rn < 2; ++rn) wc[rn] = wnx[rn];                                           \
        }                                                                                                                \
    }
    ks = 0;
    for (; ks + 4 < nk; ks += 4) {
        OB_G4_STEP(0, true)
        OB_G4_STEP(1, true)
        OB_G4_STEP(2, true)
        OB_G4_STEP(3, true)
    }
    {
        OB_G4_STEP(0, false)
        OB_G4_STEP(1, false)
        OB_G4_STEP(2, false)
        OB_G4_STEP(3, false)
    }
#undef OB_G4_STEP
#undef OB_G4_MMA
#undef OB_G4_READ

    // Epilogue.  Lane: token t4 * 32 + r32 of the tile, rows rn * 32 + 8 q + 4 kg + i (i = 0..3) of the wave's 64 for accumulator quad q.
    if (!PARTIAL && n0 + wn * 64 + 64 <= N && (N & 7) == 0 && (reinterpret_cast<size_t>(u) & 15) == 0) {
        // through the idle staging LDS (every tile read finished before the last barrier): each wave transposes its 64 x 128 tile in
        // its own 18 KB and writes whole 128-byte row segments, 8 token rows per instruction (as ob_gemm3)
        constexpr int EP = 144;
        char *ep = smem + (size_t)wave * 128 * EP;
#pragma unroll
        for (int rn = 0; rn < 2; ++rn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = rn * 32 + 8 * q + 4 * kg;
                const ob_half4 g4 = *reinterpret_cast<const ob_half4 *>(g + n0 + wn * 64 + nl);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    ob_half4 ov;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ov[i] = (_Float16)(ob_round_h(acc[rn][t4][4 * q + i]) * (float)g4[i]);   // fp16(z) (:115), * g -> fp16 (:116)
                    *reinterpret_cast<ob_half4 *>(ep + (t4 * 32 + r32) * EP + nl * 2) = ov;
                }
            }
        }
        _Float16 *ub = u + (int64_t)t0 * N + n0 + wn * 64;
        const int trows = T - t0;
        float *tsp = zp;                            // optional LayerNorm tile partials (ONEBIT_FLAG_TILE_STATS), as ob_gemm3
        const int64_t ntile64 = N >> 6;
        const int tile_idx = (n0 + wn * 64) >> 6;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = j * 8 + (lane >> 3), ch = lane & 7;
            const ob_half8 v = *reinterpret_cast<const ob_half8 *>(ep + row * EP + ch * 16);
            if (row < trows) *reinterpret_cast<ob_half8 *>(ub + (int64_t)row * N + ch * 8) = v;
            if (tsp) {
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) sm += (float)v[i];
                sm += OB_DPP_F(sm, 0xB1, 0xF);
                sm += OB_DPP_F(sm, 0x4E, 0xF);
                sm += OB_DPP_F(sm, 0x141, 0xF);
                const float mu = sm * 0.015625f;
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) m2 = __builtin_fmaf((float)v[i] - mu, (float)v[i] - mu, m2);
                m2 += OB_DPP_F(m2, 0xB1, 0xF);
                m2 += OB_DPP_F(m2, 0x4E, 0xF);
                m2 += OB_DPP_F(m2, 0x141, 0xF);
                if (ch == 0 && row < trows)
                    *reinterpret_cast<ob_float2 *>(tsp + ((int64_t)(t0 + row) * ntile64 + tile_idx) * 2) = (ob_float2){sm, m2};
            }
        }
        return;
    }
#pragma unroll
    for (int rn = 0; rn < 2; ++rn) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nb = n0 + wn * 64 + rn * 32 + 8 * q + 4 * kg;
            float gn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) gn[i] = PARTIAL ? 1.0f : (float)g[min(nb + i, N - 1)];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int t = t0 + t4 * 32 + r32;
                if (t >= T) continue;
                if (PARTIAL) {
                    if (nb + 3 < N) {
                        *reinterpret_cast<ob_float4 *>(zp + (int64_t)t * N + nb) =
                            (ob_float4){acc[rn][t4][4 * q], acc[rn][t4][4 * q + 1], acc[rn][t4][4 * q + 2], acc[rn][t4][4 * q + 3]};
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (nb + i < N) zp[(int64_t)t * N + nb + i] = acc[rn][t4][4 * q + i];
                    }
                } else {
                    _Float16 o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (_Float16)(ob_round_h(acc[rn][t4][4 * q + i]) * gn[i]);
                    if (nb + 3 < N) {
                        const ob_half4 ov = {o[0], o[1], o[2], o[3]};
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
}

template <bool PARTIAL>
__global__ __launch_bounds__(256, 2) void ob_gemm4_f16_kernel(
    const uint32_t *__restrict__ W, int64_t ldw_words, const _Float16 *__restrict__ a, int64_t lda,
    const _Float16 *__restrict__ g, _Float16 *__restrict__ u, float *__restrict__ zp, int T, int K, int N, int nbn)
{
    ob_gemm4_body<PARTIAL>(W, ldw_words, a, lda, g, u, zp, T, K, N, nbn, ob_g3_bid());
}

// grouped form: see ob_gemm3g_f16_kernel
__global__ __launch_bounds__(256, 2) void ob_gemm4g_f16_kernel(const ObG3Group G)
{
    const int bid = ob_g3_bid();
    const int p = (bid >= G.tile_end[0] ? 1 : 0) + (bid >= G.tile_end[1] ? 1 : 0);
    const int b0 = p == 0 ? 0 : (p == 1 ? G.tile_end[0] : G.tile_end[1]);
    const uint32_t *W = p == 0 ? G.W[0] : (p == 1 ? G.W[1] : G.W[2]);
    const _Float16 *a = p == 0 ? G.a[0] : (p == 1 ? G.a[1] : G.a[2]);
    const _Float16 *g = p == 0 ? G.g[0] : (p == 1 ? G.g[1] : G.g[2]);
    _Float16 *u = p == 0 ? G.u[0] : (p == 1 ? G.u[1] : G.u[2]);
    const int N = p == 0 ? G.N[0] : (p == 1 ? G.N[1] : G.N[2]);
    const int nbn = p == 0 ? G.nbn[0] : (p == 1 ? G.nbn[1] : G.nbn[2]);
    ob_gemm4_body<false>(W, G.ldw_words, a, G.lda, g, u, nullptr, G.T, G.K, N, nbn, bid - b0);
}
