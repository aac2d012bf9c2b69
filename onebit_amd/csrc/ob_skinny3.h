// Skinny 1-bit GEMM, LDS-DMA form: 2 <= T <= 64 tokens on PRE-SCALED activation rows a = fp16(x * h) (written by the
// producers: onebit_rows_res_ln_rms / onebit_rows_swiglu with h_next, the batched attention kernel), fp16 MFMA
// (v_mfma_f32_16x16x32_f16), fp32 accumulate -- the arithmetic of ob_skinny.h.
//
// What the first form (ob_skinny.h) spends its time on is not the matrix pipe: per 512-k phase a wave issues 215
// instructions for 16 MFMAs -- 4 global loads, 16 multiplies by h, LDS stores, tail selects -- between two workgroup
// barriers, and 344 workgroups each scale the same [T, K] block again.  Here
//   * the rows arrive scaled, so a piece of them goes global -> LDS by DMA (global_load_lds_dwordx4: no VGPRs, no
//     multiplies, no ds_write), and so do the packed weights;
//   * workgroup = 4 waves = 16 * RNT rows x all T tokens (RT groups of 16); the waves split K in PIECES of 128 weights
//     (wave w: pieces w, w + 4, ...), and everything a wave touches before the final sum is its own: its pieces of the
//     activation rows, its 16 bytes of every packed row, its ring of NBUF LDS buffers.  So there is NO workgroup barrier in
//     the K loop -- a wave waits for ITS piece with a counted s_waitcnt vmcnt (the younger pieces stay in flight) and the
//     four waves drift apart freely: while one waits another multiplies;
//   * a piece = two sub-tiles of 64 k in the LDS layout of ob_gemm2.h (token rows of 128 B, 16-byte chunks XOR-swizzled by
//     ob_g2_swz: operand reads bank-conflict free) + 16 B of each packed row.  K order inside a sub-tile: lane group gq owns
//     k = 16 gq .. 16 gq + 15 (MFMA sub-step S: + 8 S), so its 16 sign bits are ONE half-word of the piece's packed words
//     (one ds_read_b32 + ob_expand16 per row tile and sub-tile, feeding 2 * RT MFMAs) and its B operands are chunks
//     2 gq + S of the natural k order;
//   * per sub-tile and wave: RNT expansions (~22 VALU each), 2 * RT operand reads, 2 * RNT * RT MFMAs, 4-8 DMA
//     instructions: ~140 instructions per 16 MFMAs at RT = 2, RNT = 4.
// The four partial accumulators meet in LDS at the end (fixed order: deterministic); epilogue as ob_skinny.h:
// fp16(z) * g (bitnet.py:115-116) or fp32 partial sums (K-slices summed by the consumer), optional LayerNorm tile partials.
#pragma once
#include "ob_skinny.h"
#include "ob_gemm2.h"

// One projection (or one K-slice of one: W, a advanced to the slice, K = its length, PARTIAL output).
struct ObSk3Proj {
    const uint32_t *W; long long ldw_words;
    const _Float16 *g;
    const _Float16 *a;            // pre-scaled rows [T, lda] of THIS projection (already advanced to the K-slice)
    _Float16 *u;                  // !PARTIAL: fp16 [T, N]
    float *zp;                    // PARTIAL: fp32 sums [T, N]
    float *st;                    // !PARTIAL, optional: per-token LayerNorm tile partials (ObSkinnyProj::st)
    int N, K, wg_end;             // workgroups [previous wg_end, wg_end) of the grid
};
struct ObSk3Args {
    ObSk3Proj p[3];
    long long lda;
    int T;
    unsigned long long *dbg;      // -DOB_PROFILE_STAMPS builds: 16 cycle stamps per wave (tools/skinny3_phase_probe.py)
};

#define OB_SK3_NWD(RNT_) ((16 * (RNT_) + 63) / 64)                                   // weight DMAs per piece
#define OB_SK3_PIECE(RT_, RNT_) (2 * 16 * (RT_) * 128 + OB_SK3_NWD(RNT_) * 1024)     // LDS bytes per piece
#define OB_SK3_LDS(RT_, RNT_, NBUF_, NW_) ((NW_) * (NBUF_) * OB_SK3_PIECE(RT_, RNT_))

template <bool PARTIAL, int RT, int RNT, int NBUF, int NW>
__global__ __launch_bounds__(64 * NW) void ob_skinny3_kernel(const ObSk3Args A)
{
    constexpr int TT = 16 * RT;                 // tokens of the tile
    constexpr int ND = TT / 8;                  // DMA instructions per sub-tile (8 token rows x 128 B each)
    constexpr int NWD = OB_SK3_NWD(RNT);
    constexpr int PD = 2 * ND + NWD;            // DMA instructions per piece and wave
    constexpr int SUB = TT * 128;               // bytes per sub-tile
    constexpr int PIECE = OB_SK3_PIECE(RT, RNT);
    static_assert(NBUF >= 2 && (NBUF - 1) * PD <= 63, "vmcnt is a 6-bit counter");
    static_assert(NW * RNT * RT * 1024 <= OB_SK3_LDS(RT, RNT, NBUF, NW), "reduction buffer larger than the rings");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef OB_PROFILE_STAMPS
    unsigned long long stamp_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define OB_SK_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp_[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define OB_SK_STAMP(i) do { } while (0)
#endif
    OB_SK_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    const int pi = (int)blockIdx.x < A.p[0].wg_end ? 0 : ((int)blockIdx.x < A.p[1].wg_end ? 1 : 2);
    const ObSk3Proj P = pi == 0 ? A.p[0] : (pi == 1 ? A.p[1] : A.p[2]);
    const int wg0 = pi == 0 ? 0 : (pi == 1 ? A.p[0].wg_end : A.p[1].wg_end);
    const int T = A.T, K = P.K, N = P.N;
    const int n0 = ((int)blockIdx.x - wg0) * 16 * RNT;
    const int np = K >> 7;                      // pieces of 128 k (host-checked: K % 128 == 0, K >= 512)
    const int nw = (np - wave + NW - 1) / NW;   // this wave's pieces: wave, wave + NW, ...

    // per-lane DMA sources as 32-bit byte offsets from scalar bases (host-checked: T * lda * 2 and N * ldw * 4 below 4 GB)
    uint32_t asrc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int row = 8 * d + (lane >> 3);
        const int c = ob_g2_swz(row, lane & 7);                                     // physical chunk lane & 7 holds logical chunk c
        asrc[d] = (uint32_t)(((int64_t)min(row, T - 1) * A.lda + c * 8) * 2);
    }
    uint32_t wsrc[NWD];
#pragma unroll
    for (int j = 0; j < NWD; ++j) wsrc[j] = (uint32_t)((int64_t)min(n0 + 64 * j + lane, N - 1) * P.ldw_words * 4);
    const char *abase = reinterpret_cast<const char *>(P.a);
    const char *wbase = reinterpret_cast<const char *>(P.W);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + (uint32_t)wave * (NBUF * PIECE);
    // (issued from inline asm, M0 saved / restored inside the statement: see ob_gemm3_f16_kernel)
    auto dma16 = [&](const char *base, uint32_t voff, uint32_t lds_addr) {
        uint32_t m0_keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_keep) : "s"(lds_addr), "v"(voff), "s"(base) : "memory");
    };
    // Every workgroup needs every piece of the rows; walked in the same order by all of them, the CUs of an XCD ask the
    // same L2 lines (one channel: token rows are a multiple of 4 KB apart) at the same time.  Workgroup j of an XCD starts
    // at piece j instead (sums are over the same pieces: order of fp32 additions inside a wave changes, nothing else).
#ifndef OB_SK3_ROT
#define OB_SK3_ROT 1
#endif
    int rot = OB_SK3_ROT ? (int)(blockIdx.x >> 3) & 31 : 0;
    while (rot >= np) rot -= np;                // (np >= 4: a few scalar iterations at most; no integer division in the prologue)
    // (one wait per piece.  Measured and dropped: a separate wait per sub-tile with the transfers in use order -- packed rows
    //  first: every transfer queues behind their HBM fetch, 32-slot step 2.42 vs 2.36 ms; sub-tile 0, rows, sub-tile 1: 2.39)
    auto issue = [&](int i, int buf) {
        int piece = wave + NW * i + rot;
        piece = piece >= np ? piece - np : piece;
        const char *ab = abase + (size_t)piece * 256;                               // 128 k * 2 B
        const char *wb = wbase + (size_t)piece * 16;
        const uint32_t dst = lds0 + (uint32_t)buf * PIECE;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int d = 0; d < ND; ++d) dma16(ab + s * 128, asrc[d], dst + s * SUB + d * 1024);
#pragma unroll
        for (int j = 0; j < NWD; ++j) dma16(wb, wsrc[j], dst + 2 * SUB + j * 1024);
    };

    ob_float4 acc[RNT][RT];
#pragma unroll
    for (int a = 0; a < RNT; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) acc[a][b] = (ob_float4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < nw) issue(i, i);
    OB_SK_STAMP(1);

    const char *ring = smem + (size_t)wave * (NBUF * PIECE);
    const int bsw[2] = {ob_g2_swz(r, 2 * gq) * 16, ob_g2_swz(r, 2 * gq + 1) * 16};
    const int woff = r * 16 + 4 * (gq >> 1), wsh = 16 * (gq & 1);
    int buf = 0, nxt = NBUF - 1;                // buffer of piece i / of piece i + NBUF - 1 (= the one piece i - 1 used)
    for (int i = 0; i < nw; ++i) {
        // request piece i + NBUF - 1 BEFORE the math of piece i (every LDS read of its buffer -- piece i - 1 -- has returned:
        // the MFMAs that consumed them are issued), then wait for piece i with a COUNTED vmcnt: loads complete in order, so
        // it has landed once at most the younger pieces' transfers are outstanding
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + NBUF - 1 < nw) issue(i + NBUF - 1, nxt);
        const int rem = min(nw - 1, i + NBUF - 1) - i;
        if (NBUF >= 4 && rem >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PD <= 63 ? 3 * PD : 63) : "memory");
        else if (NBUF >= 3 && rem == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PD <= 63 ? 2 * PD : 63) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (i == 0) OB_SK_STAMP(2);
        const char *pb = ring + (size_t)buf * PIECE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t e[RNT][8];
#pragma unroll
            for (int rn = 0; rn < RNT; ++rn) {
                const uint32_t word = *reinterpret_cast<const uint32_t *>(pb + 2 * SUB + rn * 256 + woff + 8 * s);
                ob_expand16((word >> wsh) & 0xffffu, e[rn]);
            }
#pragma unroll
            for (int S = 0; S < 2; ++S) {
                ob_half8 bop[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    bop[rt] = *reinterpret_cast<const ob_half8 *>(pb + s * SUB + (rt * 16 + r) * 128 + bsw[S]);
#pragma unroll
                for (int rn = 0; rn < RNT; ++rn) {
                    const ob_u32x4 av = {e[rn][4 * S + 0], e[rn][4 * S + 1], e[rn][4 * S + 2], e[rn][4 * S + 3]};
                    const ob_half8 aop = __builtin_bit_cast(ob_half8, av);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[rn][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bop[rt], acc[rn][rt], 0, 0, 0);
                }
            }
        }
        if (i == 0) OB_SK_STAMP(3);
        nxt = buf;
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        if (i == 0) OB_SK_STAMP(4);
        if (i == 1) OB_SK_STAMP(5);
        if (i == 3) OB_SK_STAMP(6);
    }
    OB_SK_STAMP(7);
    // weight_scale of this thread's first output slot, requested before the barriers (its L2 / HBM round trip would
    // otherwise follow them)
    constexpr int NSLOT = RNT * RT * 64;
    const _Float16 *__restrict__ g = P.g;
    const bool gvec = !PARTIAL && (N & 3) == 0 && (reinterpret_cast<size_t>(g) & 7) == 0;
    ob_half4 g4p = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
    if (gvec && tid < NSLOT) {
        const int nb0 = n0 + ((tid >> 6) / RT) * 16 + 4 * ((tid & 63) >> 4);
        if (nb0 + 3 < N) g4p = *reinterpret_cast<const ob_half4 *>(g + nb0);
    }
    __syncthreads();                            // every wave is done with its ring: the memory becomes the reduction buffer

    // the waves' partial accumulators meet in LDS: [wave][rn][rt][lane] float4; output slot (rn, rt, lane) is summed by one thread
    ob_float4 *zr = reinterpret_cast<ob_float4 *>(smem);
#pragma unroll
    for (int rn = 0; rn < RNT; ++rn)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) zr[((wave * RNT + rn) * RT + rt) * 64 + lane] = acc[rn][rt];
    __syncthreads();
    OB_SK_STAMP(8);
    for (int slot = tid; slot < NSLOT; slot += 64 * NW) {                              // (uniform per wave: NSLOT % 64 == 0)
        const int sl = slot & 63, rt = (slot >> 6) % RT, rn = (slot >> 6) / RT;
        ob_float4 z = zr[((0 * RNT + rn) * RT + rt) * 64 + sl];
#pragma unroll
        for (int w = 1; w < NW; ++w) z += zr[((w * RNT + rn) * RT + rt) * 64 + sl];
        const int t = rt * 16 + (sl & 15);
        const int ntile = n0 + rn * 16;
        const int nb = ntile + 4 * (sl >> 4);
        if (PARTIAL) {
            if (t >= T) continue;
            if (nb + 3 < N && (N & 3) == 0) {
                *reinterpret_cast<ob_float4 *>(P.zp + (int64_t)t * N + nb) = z;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (nb + i < N) P.zp[(int64_t)t * N + nb + i] = z[i];
            }
            continue;
        }
        _Float16 o[4];
        float sm = 0.f;
        ob_half4 g4;
        if (gvec && nb + 3 < N) g4 = slot == tid ? g4p : *reinterpret_cast<const ob_half4 *>(g + nb);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) g4[i] = g[min(nb + i, N - 1)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = (_Float16)(ob_round_h(z[i]) * (float)g4[i]);                     // fp16(z) (bitnet.py:115), * g -> fp16 (:116)
            sm += (float)o[i];
        }
        if (P.st) {                                                                 // (uniform per workgroup)
            // the 16 rows of the tile for token t live in lanes sl, sl ^ 16, sl ^ 32, sl ^ 48 (4 rows each)
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            const float mu = sm * 0.0625f;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) m2 = __builtin_fmaf((float)o[i] - mu, (float)o[i] - mu, m2);
            m2 += __shfl_xor(m2, 16);
            m2 += __shfl_xor(m2, 32);
            if (sl < 16 && t < T && ntile < N) {
                float *d = P.st + (size_t)t * ob_tile_stats_floats(N) + (size_t)(ntile >> 4) * 2;
                d[0] = sm; d[1] = m2;
            }
        }
        if (t >= T) continue;
        if (nb + 3 < N && (N & 3) == 0) {
            const ob_half4 ov = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<ob_half4 *>(P.u + (int64_t)t * N + nb) = ov;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (nb + i < N) P.u[(int64_t)t * N + nb + i] = o[i];
        }
    }
#ifdef OB_PROFILE_STAMPS
    OB_SK_STAMP(9);
    if (A.dbg && lane == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i_ = 0; i_ < 16; ++i_) A.dbg[(blockIdx.x * 8 + wave) * 16 + i_] = stamp_[i_];
    }
#endif
#undef OB_SK_STAMP
}
