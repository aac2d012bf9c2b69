// Skinny 1-bit GEMM, second form: 2 <= T <= 32 tokens on PRE-SCALED activation rows a = fp16(x * h) (written by the
// producers: onebit_rows_res_ln_rms / onebit_rows_swiglu with h_next, the batched attention kernel).  Used where it
// measured faster than the first form (ob_skinny.h): ONE projection with ONE 512-weight chunk per wave (K <= 4096) -- the
// batched decode step's o_proj (7.9 vs 10.6 us at 7B, 32 slots) and ONEBIT_FLAG_PRESCALED calls of short prompts.
//
// The first form walks K in 512-element phases through LDS with a workgroup barrier per phase (8 phases of ~2400 cycles
// for K = 4096 whatever the matrix pipe does).  This one is the decode GEMV (ob_decode.h) generalised to a tile of tokens:
//   * persistent grid, one 512-thread workgroup per CU; workgroup b owns 16-row tiles b, b + G, ... of the projection;
//   * its 8 waves split K in 512-weight chunks (wave w: chunk w): a lane's packed words go global -> VGPR -> MFMA A
//     operand (one 16-byte non-temporal load per tile, all issued at kernel entry), NO barrier until the partial
//     accumulators meet at the end;
//   * the activations of the chunk are the B operands, held in registers for every row tile; they come from the
//     L2-resident rows in whole 128-byte lines through a WAVE-PRIVATE LDS image (half a chunk at a time, no barrier: a
//     wave's LDS operations are ordered), the second half-chunk's lines in flight under the first one's MFMAs;
//   * with the scaling done by the producer the only VALU work is the sign expansion (17 instructions per 4 MFMAs at
//     T = 32).
// The template is general in (projections, tile slots, chunks per wave); with several projections or chunks per wave a
// workgroup pulls one [T, 512-chunk] slab per (projection, chunk) with half a chunk of lookahead and every projection
// has its OWN scaled rows -- q|k|v 22.4, gate|up 18.1, down 16.7 us against 14.8 / 14.8 / 10.6 for the first form -- so
// only the single-set instances are built (onebit_hip.hip, ob_launch_skinny2).
// Epilogue: partial accumulators of the 8 waves through LDS, fixed-order fp32 sum, fp16(z) * g (bitnet.py:115-116),
// optional per-token LayerNorm tile partials for the consumer (as ob_skinny.h).
#pragma once
#include "ob_decode.h"
#include "ob_skinny.h"

struct ObSk2Proj {
    const uint32_t *W; long long ldw_words;
    const _Float16 *g;
    const _Float16 *a;            // pre-scaled activations [T, K] (row pitch ObSk2Args::lda)
    _Float16 *u;                  // out [T, N]
    float *st;                    // optional per-token tile partials (ob_tile_stats_floats(N) floats per token), N % 16 == 0
    int N;
};
struct ObSk2Args {
    ObSk2Proj p[3];
    long long lda;
    int T, K;
};

template <int TT, int MS, int NPROJ, int KV>
__global__ __launch_bounds__(512) void ob_skinny2_kernel(const ObSk2Args A)
{
    constexpr int MT = MS * NPROJ, NSET = NPROJ * KV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, gq = lane >> 4;
    const int G = gridDim.x, T = A.T, K = A.K;
    const int nchunks = (K + 511) >> 9;
    const ObSk2Proj PP[3] = {A.p[0], A.p[NPROJ > 1 ? 1 : 0], A.p[NPROJ > 2 ? 2 : 0]};

    int trow[MT];
    bool tval[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int ti = (j / NPROJ) * G + (int)blockIdx.x;
        tval[j] = ti < ((PP[j % NPROJ].N + 15) >> 4);
        trow[j] = (tval[j] ? ti : 0) << 4;
    }
    // ---- every packed word of the workgroup, in use order (non-temporal: read once per step)
    ob_u32x4 wreg[MT][KV];
#pragma unroll
    for (int p = 0; p < NPROJ; ++p)
#pragma unroll
        for (int ci = 0; ci < KV; ++ci)
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                const int j = s * NPROJ + p;
                wreg[j][ci] = ob_dec_load_w<true>(PP[p].W, PP[p].N, K, (int)PP[p].ldw_words, trow[j], min(wave + ci * OB_DEC_WAVES, nchunks - 1), lane);
            }
    // ---- B fragments.  MFMA step (q, hf, s2) of a chunk multiplies k = chunk * 512 + gq * 128 + q * 32 + 16 hf + 8 s2 .. + 7
    //      (the 8 weights one quarter of an expanded half-word covers); lane (token = 16 tt + r, k-group gq) needs those 8
    //      halves of its token row.  Loading them directly (16 rows x 4 scattered 16-byte pieces per instruction) drowns
    //      the texture addresser: 64 requests per instruction for 16 useful bytes each (measured: the 32-slot step 5.8 ms).
    //      So a HALF-SET (words q = 2 hh, 2 hh + 1: per token row the four 128-byte segments gq) is fetched with whole
    //      128-byte lines (8 lanes per segment), written to a WAVE-PRIVATE LDS image and read back as fragments -- LDS
    //      operations of one wave complete in order, so no barrier is involved; the next half-set's lines are requested
    //      before this one's MFMAs and land underneath them.  Image: [row][gq][8 pieces of 16 B], piece index XORed with
    //      row & 7 (rows are 512 B apart: unswizzled, the 16 rows of a fragment read would share one bank group).
    constexpr int ROWS = 16 * TT, NLD = ROWS / 2;            // 16-byte pieces per lane and half-set
    char *wl = smem + (size_t)wave * ROWS * 512;
    ob_u32x4 stg[NLD];
    auto half_src_ok = [&](int set, int hh, int i, int &row, int &seg, int &pc, int64_t &off) -> bool {
        const int ci = set % KV;
        const int chunk = wave + ci * OB_DEC_WAVES;
        const int pid = i * 64 + lane;
        row = pid >> 5; seg = (pid >> 3) & 3; pc = pid & 7;
        const int k = chunk * 512 + seg * 128 + hh * 64 + pc * 8;
        const bool ok = chunk < nchunks && k < K;
        off = (int64_t)min(row, T - 1) * A.lda + (ok ? k : 0);
        return ok;
    };
    auto issue_half = [&](int set, int hh) {
        const _Float16 *ap = PP[set / KV].a;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int row, seg, pc; int64_t off;
            (void)half_src_ok(set, hh, i, row, seg, pc, off);
            stg[i] = *reinterpret_cast<const ob_u32x4 *>(ap + off);
        }
    };
    auto to_lds = [&](int set, int hh) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int row, seg, pc; int64_t off;
            const bool ok = half_src_ok(set, hh, i, row, seg, pc, off);
            const ob_u32x4 v = ok ? stg[i] : (ob_u32x4){0u, 0u, 0u, 0u};        // beyond K: sign bits 0 would read as +1
            *reinterpret_cast<ob_u32x4 *>(wl + (((row * 4 + seg) * 8 + (pc ^ (row & 7))) << 4)) = v;
        }
    };
    ob_half8 bfr[8][TT];
    auto read_frags = [&]() {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const int row = 16 * tt + r;
                bfr[pc][tt] = *reinterpret_cast<const ob_half8 *>(wl + (((row * 4 + gq) * 8 + (pc ^ (row & 7))) << 4));
            }
    };
    issue_half(0, 0);

    ob_float4 acc[MT][TT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[j][tt] = (ob_float4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int hs = 0; hs < 2 * NSET; ++hs) {
        const int set = hs >> 1, hh = hs & 1;
        const int p = set / KV, ci = set % KV;
        const bool live = wave + ci * OB_DEC_WAVES < nchunks;                          // wave-uniform
        to_lds(set, hh);
        read_frags();
        if (hs + 1 < 2 * NSET) issue_half((hs + 1) >> 1, (hs + 1) & 1);               // lands underneath the MFMAs below
        if (live) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                    for (int s = 0; s < MS; ++s) {
                        const int j = s * NPROJ + p;
                        if (!tval[j]) continue;                                        // workgroup-uniform
                        uint32_t e[8];
                        ob_expand16((wreg[j][ci][2 * hh + q2] >> (16 * hf)) & 0xffffu, e);
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            ob_u32x4 av = {e[4 * s2 + 0], e[4 * s2 + 1], e[4 * s2 + 2], e[4 * s2 + 3]};
                            ob_half8 aop;
                            __builtin_memcpy(&aop, &av, 16);
#pragma unroll
                            for (int tt = 0; tt < TT; ++tt)
                                acc[j][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bfr[q2 * 4 + hf * 2 + s2][tt], acc[j][tt], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();                            // every wave is done with its staging image: the memory becomes the reduction buffer

    // ---- the 8 waves' partial accumulators meet in LDS: [wave][j][tt][lane] float4
    ob_float4 *zr = reinterpret_cast<ob_float4 *>(smem);
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) zr[((wave * MT + j) * TT + tt) * 64 + lane] = acc[j][tt];
    __syncthreads();
    constexpr int NSLOT = MT * TT * 64;
    for (int slot = tid; slot < NSLOT; slot += 512) {                                  // (uniform trip count per wave: NSLOT % 64 == 0)
        const int sl = slot & 63, tt = (slot >> 6) % TT, j = (slot >> 6) / TT;
        ob_float4 z = zr[((0 * MT + j) * TT + tt) * 64 + sl];
#pragma unroll
        for (int w = 1; w < OB_DEC_WAVES; ++w) z += zr[((w * MT + j) * TT + tt) * 64 + sl];
        const int p = j % NPROJ;
        // projection-dependent pointers by compile-time-bounded selects
        const ObSk2Proj P = p == 0 ? PP[0] : (p == 1 ? PP[1] : PP[2]);
        // slot-dependent tile: recompute (trow[] is a register array indexed statically only)
        const int ti = (j / NPROJ) * G + (int)blockIdx.x;
        const bool tv = ti < ((P.N + 15) >> 4);
        const int t = tt * 16 + (sl & 15);
        const int nb = (ti << 4) + 4 * (sl >> 4);
        _Float16 o[4];
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gn = (float)P.g[min(nb + i, P.N - 1)];
            o[i] = (_Float16)(ob_round_h(z[i]) * gn);                                  // fp16(z) (bitnet.py:115), * g -> fp16 (:116)
            sm += (float)o[i];
        }
        if (P.st) {                                                                    // (uniform per wave: j is)
            // the 16 rows of the tile for token t live in lanes sl, sl ^ 16, sl ^ 32, sl ^ 48 (4 rows each)
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            const float mu = sm * 0.0625f;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) m2 = __builtin_fmaf((float)o[i] - mu, (float)o[i] - mu, m2);
            m2 += __shfl_xor(m2, 16);
            m2 += __shfl_xor(m2, 32);
            if (sl < 16 && t < T && tv) {
                float *d = P.st + (size_t)t * ob_tile_stats_floats(P.N) + (size_t)ti * 2;
                d[0] = sm; d[1] = m2;
            }
        }
        if (t >= T || !tv) continue;
        if (nb + 3 < P.N && (P.N & 3) == 0) {
            ob_half4 ov = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<ob_half4 *>(P.u + (int64_t)t * P.N + nb) = ov;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (nb + i < P.N) P.u[(int64_t)t * P.N + nb + i] = o[i];
        }
    }
}
