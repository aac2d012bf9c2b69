// Prefill 1-bit GEMM, large-T form (gfx950): z[t][n] = sum_k s[n][k] * fp16(x[t][k] * h[k]) on
// v_mfma_f32_16x16x32_f16, fp32 accumulate (bitnet.py:113-115); epilogue fp16(z) * g -> fp16 (:115-116).
//
// Workgroup tile 256 rows (n) x 256 tokens (t), 8 waves as 4 (n) x 2 (t); a wave owns 64 rows x 128
// tokens = 4 x 8 MFMA tiles (128 accumulator VGPRs).  What the shape buys over the 128 x 128 / 4-wave
// kernel (ob_gemm.h): the packed signs of a row tile are expanded to +-1.0 fp16 ONCE per K step and
// feed 8 token tiles instead of 4 -- the expansion (VALU, ~1.1 op per weight) was as long as the MFMA
// work it fed; here it is half of it, and the two waves of a SIMD cover each other.
// K advances 64 per step through THREE LDS activation buffers with ONE barrier per step:
//   step ks:  global loads of tile ks + 2 -> registers (issued first, land under the MFMAs)
//             64 MFMAs per wave on buffer ks % 3
//             registers -> (x * h: one v_pk_mul_f16 per pair = the fp16 rounding of bitnet.py:113) ->
//             buffer (ks + 2) % 3, last read in step ks - 1, i.e. before the barrier every wave has passed
//             barrier
// Weights never touch LDS: a lane's packed dword (its row, 32 of the step's 64 k) goes global -> VGPR
// two steps ahead and is expanded in registers (ob_expand16).  Activation rows are 128 B with their
// 16-byte chunks XOR-swizzled: operand reads and staging stores are both bank-conflict free.
// Workgroups are renumbered so that consecutive ids on one XCD share the token tile (L2 reuse).
#pragma once
#include "ob_common.h"

#define OB_G2_N 256
#define OB_G2_T 256
#define OB_G2_K 64
#define OB_G2_PITCH 64       // halves per LDS row: 128 B, 16-byte chunks XOR-swizzled (see ob_g2_swz)
#define OB_G2_THREADS 512
#define RN 4
#define RT 8
#define OB_G2_LDS (3 * OB_G2_T * OB_G2_PITCH * 2)

// LDS image of an activation tile: row t (128 B = 8 chunks of 8 halves), chunk c stored at position
// c ^ f(t), f(t) = bit 1 of t | (t & 4).  With this f the 16 lanes of every ds_read_b128 lane group
// ({0-3, 12-15, 20-27}, ...: 8 rows of k-group gq plus 8 rows of k-group gq + 1) cover the 64 banks
// exactly once (exhaustive search over the linear swizzles, tools note in DESIGN.md); rows written by 8
// consecutive lanes stay one contiguous 128-byte segment, so the stores are conflict-free as well.
__device__ __forceinline__ int ob_g2_swz(int t, int c) { return c ^ (((t >> 1) & 1) | (t & 4)); }

template <bool PARTIAL>
__global__ __launch_bounds__(OB_G2_THREADS, 2) void ob_gemm2_f16_kernel(
    const uint32_t *__restrict__ W, int64_t ldw_words, const _Float16 *__restrict__ x, int64_t ldx,
    const _Float16 *__restrict__ h, const _Float16 *__restrict__ g, _Float16 *__restrict__ u,
    float *__restrict__ zp, int T, int K, int N, int nbn)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 (*As)[OB_G2_T][OB_G2_PITCH] = reinterpret_cast<_Float16 (*)[OB_G2_T][OB_G2_PITCH]>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 3, wt = wave >> 2;
    const int r = lane & 15, gq = lane >> 4;

    // XCD-aware renumbering (bijective for any grid size): XCD x owns a contiguous id range
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, rem = nwg & 7;
    const int bid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (orig >> 3);
    const int tt = bid / nbn, tn = bid - tt * nbn;          // n fastest: neighbours share the token tile
    const int n0 = tn * OB_G2_N, t0 = tt * OB_G2_T;
    const int nk = K / OB_G2_K;                              // K % 64 == 0 (host-checked)

    // staging: thread -> (token st_t + 64 i, halves st_k .. st_k + 7); 8 lanes cover one 128-byte row segment
    const int st_t = tid >> 3, st_k = (tid & 7) * 8;
    const _Float16 *xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xrow[i] = x + (int64_t)min(t0 + st_t + 64 * i, T - 1) * ldx + st_k;
    const _Float16 *hp = h + st_k;
    const int st_sw = ob_g2_swz(st_t, tid & 7) * 8;           // (st_t + 64 i) has the same low bits as st_t

    // weights: this lane's row of each of the wave's 4 row tiles; word (2 ks + (gq >> 1)), half (gq & 1)
    const uint32_t *wrow[RN];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) wrow[rn] = W + (int64_t)min(n0 + wn * 64 + rn * 16 + r, N - 1) * ldw_words + (gq >> 1);
    const int wsh = (gq & 1) * 16;

    ob_float4 acc[RN][RT];
#pragma unroll
    for (int a = 0; a < RN; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) acc[a][b] = (ob_float4){0.f, 0.f, 0.f, 0.f};

    ob_half8 xs[4], hs;
    uint32_t w0[RN], w1[RN];             // packed words of steps ks (current) and ks + 1

    auto load_x = [&](int ks) {
        hs = *reinterpret_cast<const ob_half8 *>(hp + ks * OB_G2_K);
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[i] = *reinterpret_cast<const ob_half8 *>(xrow[i] + ks * OB_G2_K);
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<ob_half8 *>(&As[buf][st_t + 64 * i][st_sw]) = xs[i] * hs;           // fp16(x * h)
    };
    auto load_w = [&](int ks, uint32_t (&w)[RN]) {
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) w[rn] = wrow[rn][2 * ks];
    };

    // prologue: tiles 0 and 1 staged, weights of steps 0 and 1 in registers
    load_x(0);
    load_w(0, w0);
    store_x(0);
    if (nk > 1) { load_x(1); load_w(1, w1); store_x(1); } else { load_w(0, w1); }
    // a compiler-visible use of every loop-carried register that a prologue load defined: the compiler
    // waits for those loads HERE, and carries no pending-load state into the loop (where its counted
    // waits would land on the asm loads of the steady state)
    asm volatile("" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]), "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]));

    // Steady state.  The global loads of tile ks + 2 are issued at the top of step ks and consumed after
    // its MFMA block.  They are issued from inline asm: hipcc's s_waitcnt insertion is conservative
    // across the loop back-edge (it waited for a tile's loads at the TOP of the MFMA block -- a memory
    // round trip per step with the matrix pipe idle: 784 -> 957 TFLOP/s); the compiler does not count
    // asm loads, and the one wait they need (vmcnt(0), naming every destination register) sits after
    // the MFMA block.  (A second register set -- tile ks + 3 in flight across two MFMA blocks -- measured
    // no faster and pushed the kernel into scratch, which asm-loaded registers must never see.)
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks % 3;
#if defined(OB_GEMM_ABL) && (OB_GEMM_ABL & 4)
        const bool more2 = false;                                 // ablation: no staging after the prologue
#else
        const bool more2 = ks + 2 < nk;
#endif
        ob_u32x4 xq[4], hq;
        uint32_t w2[RN] = {0u, 0u, 0u, 0u};
        if (more2) {
            const int kofs = (ks + 2) * OB_G2_K;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hq) : "v"(hp + kofs) : "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq[i]) : "v"(xrow[i] + kofs) : "memory");
#pragma unroll
            for (int rn = 0; rn < RN; ++rn) asm volatile("global_load_dword %0, %1, off" : "=v"(w2[rn]) : "v"(wrow[rn] + 2 * (ks + 2)) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        // all sign expansions of the step first (one VALU burst), then two dense bursts of 32 MFMAs: an
        // MFMA that waits for an expansion issued just before it stalls the wave's whole in-order stream
        uint32_t e[RN][8];
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
#if defined(OB_GEMM_ABL) && (OB_GEMM_ABL & 1)
            _Pragma("unroll") for (int i = 0; i < 8; ++i) e[rn][i] = w0[rn] + i;                           // ablation: no sign expansion
#else
            ob_expand16((w0[rn] >> wsh) & 0xffffu, e[rn]);
#endif
        }
#ifdef OB_G2_BURST
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            ob_half8 bop[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#if defined(OB_GEMM_ABL) && (OB_GEMM_ABL & 2)
                bop[rt] = __builtin_bit_cast(ob_half8, (ob_u32x4){w0[0] + rt, w0[1], w0[2] + s, w0[3]});     // ablation: no LDS operand reads
#else
                bop[rt] = *reinterpret_cast<const ob_half8 *>(&As[cur][wt * 128 + rt * 16 + r][ob_g2_swz(r, gq * 2 + s) * 8]);
#endif
#ifdef OB_G2_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int rn = 0; rn < RN; ++rn) {
                const ob_u32x4 av = {e[rn][4 * s + 0], e[rn][4 * s + 1], e[rn][4 * s + 2], e[rn][4 * s + 3]};
                const ob_half8 aop = __builtin_bit_cast(ob_half8, av);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rn][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bop[rt], acc[rn][rt], 0, 0, 0);
            }
#ifdef OB_G2_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#ifdef OB_G2_BURST
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more2) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(hq), "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]),
                         "+v"(w2[0]), "+v"(w2[1]), "+v"(w2[2]), "+v"(w2[3]) :: "memory");
            hs = __builtin_bit_cast(ob_half8, hq);
#pragma unroll
            for (int i = 0; i < 4; ++i) xs[i] = __builtin_bit_cast(ob_half8, xq[i]);
            store_x((ks + 2) % 3);
        }
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) { w0[rn] = w1[rn]; w1[rn] = w2[rn]; }
#if !(defined(OB_GEMM_ABL) && (OB_GEMM_ABL & 8))
        __syncthreads();
#endif
    }

    // epilogue: D[n][t]: lane holds n = 4 gq + i (i = 0..3), t = r of each 16 x 16 tile
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) {
        const int nb = n0 + wn * 64 + rn * 16 + 4 * gq;
        float gn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gn[i] = PARTIAL ? 1.0f : (float)g[min(nb + i, N - 1)];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int t = t0 + wt * 128 + rt * 16 + r;
            if (t >= T) continue;
            if (PARTIAL) {
                if (nb + 3 < N) {
                    *reinterpret_cast<ob_float4 *>(zp + (int64_t)t * N + nb) = acc[rn][rt];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (nb + i < N) zp[(int64_t)t * N + nb + i] = acc[rn][rt][i];
                }
            } else {
                _Float16 o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)(ob_round_h(acc[rn][rt][i]) * gn[i]);   // fp16(z) (:115), * g -> fp16 (:116)
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
#undef RN
#undef RT
