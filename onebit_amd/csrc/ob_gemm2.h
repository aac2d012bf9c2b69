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

// ---------------------------------------------------------------------------------------------
// LDS-DMA form: the same tiling on PRE-SCALED activations a[t][k] = fp16(x[t][k] * h[k]) (one
// elementwise pass per call, ob_scale_rows_kernel: 2 * T * K * 2 bytes of traffic, ~3 % of the GEMM at
// T = 16384).  With the rounding of bitnet.py:113 done up front a tile needs no arithmetic on its
// way in, so it goes global -> LDS by global_load_lds_dwordx4 (16 bytes per lane, 1 KB per wave
// instruction, no VGPRs, no v_pk_mul, no ds_write): staging shrinks from 9 loads + 16 multiplies +
// 4 LDS stores + a counted wait per thread and step to 4 DMA instructions per wave, which is what the
// ablations of the register-staged kernel said it costs (957 -> 1372 TFLOP/s without staging).
// The DMA writes wave-uniform base + 16 * lane: lane l of DMA i of wave w fills row (4w + i) * 8 +
// (l >> 3), position l & 7, and FETCHES chunk (l & 7) ^ f(row) -- the XOR swizzle lives in the source
// address.  Three LDS buffers, one __syncthreads per step (hipcc drains vmcnt before it: the tile
// of step ks + 2 has the whole MFMA block of step ks to land).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ob_scale_rows_kernel(const _Float16 *__restrict__ x, int64_t ldx, const _Float16 *__restrict__ h,
                                                            _Float16 *__restrict__ a, int64_t T, int K)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;      // 8 halves per thread; K % 8 == 0
    if (i >= T * K) return;
    const int64_t t = i / K;
    const int k = (int)(i - t * K);
    const ob_half8 xv = *reinterpret_cast<const ob_half8 *>(x + t * ldx + k);
    const ob_half8 hv = *reinterpret_cast<const ob_half8 *>(h + k);
    *reinterpret_cast<ob_half8 *>(a + i) = xv * hv;                       // fp16(x * h), bitnet.py:113
}

#define OB_G3_BUFS 4
#define OB_G3_LDS (OB_G3_BUFS * OB_G2_T * OB_G2_PITCH * 2 + 2 * 8192)      // 4 activation tiles + 2 weight quads
#define OB_G3_LDS_W(WT_) (OB_G3_BUFS * 128 * (WT_) * OB_G2_PITCH * 2 + 2 * 8192)
// WT = token-side waves: 2 = the 256 x 256 / 8-wave workgroup described above (one per CU: 147 KB of LDS); 1 = 256 rows x
// 128 tokens / 4 waves (80 KB: TWO workgroups per CU, each with its own barrier -- while one waits the other multiplies;
// the same wave tile, the same activation traffic per flop, twice the (tiny) weight traffic).
// logical tile id of this workgroup: consecutive ids on one XCD (see ob_flash.h)
__device__ __forceinline__ int ob_g3_bid()
{
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, rem = nwg & 7;
    return (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (orig >> 3);
}

template <bool PARTIAL, int WT>
__device__ __forceinline__ void ob_gemm3_body(
    const uint32_t *__restrict__ W, int64_t ldw_words, const _Float16 *__restrict__ a, int64_t lda,
    const _Float16 *__restrict__ g, _Float16 *__restrict__ u, float *__restrict__ zp, int T, int K, int N, int nbn, const int bid)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TTILE = 128 * WT;              // tokens per workgroup tile
    _Float16 (*As)[TTILE][OB_G2_PITCH] = reinterpret_cast<_Float16 (*)[TTILE][OB_G2_PITCH]>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 3, wt = wave >> 2;
    const int r = lane & 15, gq = lane >> 4;
    const int tt = bid / nbn, tn = bid - tt * nbn;
    const int n0 = tn * OB_G2_N, t0 = tt * TTILE;
    const int nk = K / OB_G2_K;                 // K % 256 == 0 here (host-checked): whole quads of steps

    // DMA source of this lane for each of the wave's 4 instructions per tile, as a 32-bit BYTE offset from
    // the (scalar) base pointer: global_load_lds v_off, s[base:base+1] -- half the address registers of
    // 64-bit per-lane pointers (the host checks T * lda * 2 < 4 GB)
    uint32_t src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        // k order of a step: chunk c (= 2 * k-group + sub-step) of step 2m + par covers
        // k = 128 m + 64 (c >> 2) + 32 par + 8 (c & 3) .. + 7 -- so that a lane's packed words of the two steps
        // of a pair are ADJACENT (words 4m + 2 (gq >> 1) + par): one 8-byte load per row tile and pair
        const int c = ob_g2_swz(row, lane & 7);
        src[i] = (uint32_t)(((int64_t)min(t0 + row, T - 1) * lda + (c >> 2) * 64 + (c & 3) * 8) * 2);
    }
    const char *abase = reinterpret_cast<const char *>(a);
    // The DMA is issued from inline asm: hipcc models the builtin as a FLAT access to both memory and LDS and,
    // while one is pending, degrades every wait it inserts itself -- lgkmcnt for the operand reads included --
    // to 0, which serialises the software-pipelined reads below with the MFMAs.  (M0 = LDS byte address of
    // the wave's 1 KB destination; dynamic LDS starts at this kernel's LDS offset of smem.)
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    auto dma16 = [&](const char *base, uint32_t voff, uint32_t lds_addr) {
        // M0 is saved and restored INSIDE the statement (scalar temporary), so no reserved register is
        // clobbered from the compiler's point of view: whatever it may keep in M0 across this point survives
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
    // Weights go through LDS as well: while LDS-DMA transfers are pending hipcc turns EVERY wait for an
    // ordinary global load into vmcnt(0) (and an asm load's destination registers are fair game for its
    // register allocator while the data is still in flight -- it reused them as DMA addresses), so the loop
    // has no register-destination global load at all.  One DMA instruction per wave and QUAD of steps
    // (256 k = 32 bytes of a packed row): lane l of wave w fetches row 32 w + (l & 31), 16-byte half l >> 5
    // (= pair of steps 2 q + (l >> 5)), landing at w * 1024 + (l >> 5) * 512 + (l & 31) * 16 of the quad's
    // 8 KB buffer (two buffers).  A lane's 16 sign bits of a step sit in word 2 (gq >> 1) + par of the pair's
    // 16 bytes (k order above): one ds_read_b64 per row tile and PAIR of steps, conflict-free (16 rows x 16 B,
    // the two halves of a lane group pair broadcast).
    const char *wbase = reinterpret_cast<const char *>(W);
    const uint32_t wsrc = (uint32_t)((int64_t)min(n0 + 32 * wave + (lane & 31), N - 1) * ldw_words * 4 + 16 * (lane >> 5));
    // (4-wave form: a wave fetches two 32-row blocks per quad, its own and the one 4 waves up)
    const uint32_t wsrc2 = (uint32_t)((int64_t)min(n0 + 32 * (wave + 4) + (lane & 31), N - 1) * ldw_words * 4 + 16 * (lane >> 5));
    char *wlds = smem + OB_G3_BUFS * TTILE * OB_G2_PITCH * 2;
    auto wdma = [&](int quad) {
        dma16(wbase, wsrc + 32u * (uint32_t)quad, lds0 + OB_G3_BUFS * TTILE * OB_G2_PITCH * 2 + ((uint32_t)quad & 1) * 8192 + wave_u * 1024);
        if (WT == 1)
            dma16(wbase, wsrc2 + 32u * (uint32_t)quad, lds0 + OB_G3_BUFS * TTILE * OB_G2_PITCH * 2 + ((uint32_t)quad & 1) * 8192 + (wave_u + 4) * 1024);
    };
    // row wn * 64 + rn * 16 + r: block (row >> 5) = 2 wn + (rn >> 1), row-in-block (rn & 1) * 16 + r
    const char *wrd = wlds + wn * 2048 + r * 16 + 8 * (gq >> 1);
    const int wsh = (gq & 1) * 16;

    ob_float4 acc[RN][RT];
#pragma unroll
    for (int x_ = 0; x_ < RN; ++x_)
#pragma unroll
        for (int y_ = 0; y_ < RT; ++y_) acc[x_][y_] = (ob_float4){0.f, 0.f, 0.f, 0.f};
    // the weight quad and all four tile buffers in flight before the first MFMA
    wdma(0);
    dma(0);
    dma(1);
    dma(2);
    dma(3);                                     // nk % 4 == 0, nk >= 4 (host-checked)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (drains everything once)
    ob_u32x2 wc[RN], wnx[RN];                   // packed words of the current / next pair of steps
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) {
        wc[rn] = *reinterpret_cast<const ob_u32x2 *>(wrd + (rn >> 1) * 1024 + (rn & 1) * 256);
        wnx[rn] = wc[rn];
    }
    // operand reads are software-pipelined one half-step ahead of the MFMAs that consume them (two register
    // sets): after a barrier all 8 waves would otherwise ask the LDS for 8 KB each at once and the matrix
    // pipe idles until the first answers arrive -- 40 % of the kernel without any staging, by ablation
    ob_half8 bopA[RT], bopB[RT];
    // s_setprio(1) around the MFMA clusters (two 4-wave workgroups share a CU: the wave that is multiplying keeps the issue
    // port against the other workgroup's staging / expansion instructions): +1.5 % on all three 7B shapes at T = 16384
    // (1359 / 1402 / 1258 -> 1380 / 1419 / 1270 TFLOP/s, same box, alternating runs); -DOB_G3_PRIO=0: A/B build.
    // Measured and dropped: expanding the next row tile's operand into a second register quad between the current
    // tile's MFMAs (hipcc forms every operand in ONE quad: 8 MFMAs, 8 VALU, hazard nop) -- the kernel sits at 256 VGPRs,
    // the second quad spills (1321 vs 1370 TFLOP/s).
#ifndef OB_G3_PRIO
#define OB_G3_PRIO 1
#endif
#ifndef OB_G3_ABL
#define OB_G3_ABL 0                             // timing experiments only (tools/prefill_probe.py): 1 no operand reads,
#endif                                          // 2 no sign expansion, 4 no DMA, 8 no wait / barrier -- results are wrong
#define OB_G3_READ(DST, BUF, S)                                                                                          \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                    \
        if (!(OB_G3_ABL & 1) || ks < 0)                                                                                  \
            DST[rt] = *reinterpret_cast<const ob_half8 *>(&As[BUF][wt * 128 + rt * 16 + r][ob_g2_swz(r, gq * 2 + (S)) * 8]);
    // sign expansion (ob_expand16's arithmetic) done PER operand, right before its 8 MFMAs: 4 live registers
    // instead of 32 for the whole step
#define OB_G3_MMA(SRC, S)                                                                                                \
    _Pragma("unroll") for (int rn = 0; rn < RN; ++rn) {                                                                  \
        ob_u32x4 av;                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                                    \
            av[p] = (OB_G3_ABL & 2) ? cw[rn] + p : ((cw[rn] << (15 - 2 * (4 * (S) + p))) & mask) | (0x3C003C00u & ~mask);  \
        const ob_half8 aop = __builtin_bit_cast(ob_half8, av);                                                           \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                \
            acc[rn][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, SRC[rt], acc[rn][rt], 0, 0, 0);                    \
    }
    int ks = -1;
    OB_G3_READ(bopA, 0, 0)
    if (OB_G3_ABL & 1) { OB_G3_READ(bopB, 0, 1) }

    // One K step as a macro: Q (the step within its quad) and STEADY are literals, so the steady-state loop
    // has NO branch in it.  Step kc:
    //   read the second half's operands | 32 MFMAs of the first half | COUNTED wait + raw barrier (tile
    //   kc + 1 has landed for everyone, every read of tile kc is complete; a __syncthreads would drain the
    //   younger transfers) | [Q == 0: weights of the next quad] DMA of tile kc + 4 into tile kc's buffer |
    //   [Q odd: ds_read the next pair's packed words] read the FIRST half of step kc + 1 | 32 MFMAs of the
    //   second half.
    // In-order completion: tile kc + 1 (requested in step kc - 3) has landed once at most the transfers of
    // steps kc - 2 and kc - 1 are outstanding: 4 + 4 + one weight DMA if either opens a quad (Q = 1, 2) --
    // and everything older, including the next quad's weights (step 4j, first read in step 4j + 3), with it.
#define OB_G3_STEP(Q, STEADY)                                                                                            \
    {                                                                                                                    \
        const int kc = ks + (Q), cur = kc & (OB_G3_BUFS - 1);                                                            \
        uint32_t cw[RN];                                                                                                 \
        _Pragma("unroll") for (int rn = 0; rn < RN; ++rn) {                                                              \
            const uint32_t b16 = (wc[rn][(Q) & 1] >> wsh) & 0xffffu;                                                     \
            cw[rn] = (b16 & 0x5555u) | ((b16 & 0xAAAAu) << 15);                                                          \
        }                                                                                                                \
        uint32_t mask = 0x80008000u;                                                                                     \
        asm("" : "+s"(mask));                                                                                            \
        OB_G3_READ(bopB, cur, 1)                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (OB_G3_PRIO) __builtin_amdgcn_s_setprio(1);                                                                   \
        OB_G3_MMA(bopA, 0)                                                                                               \
        if (OB_G3_PRIO) __builtin_amdgcn_s_setprio(0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (OB_G3_ABL & 8) {                                                                                             \
        } else if (STEADY) {                                                                                             \
            if (((Q) == 1 || (Q) == 2) && WT == 1) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
            else if ((Q) == 1 || (Q) == 2) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");      \
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                \
        } else {                                                                                                         \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                     \
        }                                                                                                                \
        if ((STEADY) && !(OB_G3_ABL & 4)) {                                                                              \
            if ((Q) == 0) wdma((kc >> 2) + 1);                                                                           \
            dma(kc + 4);                                                                                                 \
        }                                                                                                                \
        if (((Q) & 1) && ((STEADY) || kc + 1 < nk)) {                                                                    \
            const int pm = (kc + 1) >> 1;                                                                                \
            const char *wq = wrd + ((pm >> 1) & 1) * 8192 + (pm & 1) * 512;                                              \
            _Pragma("unroll") for (int rn = 0; rn < RN; ++rn)                                                            \
                wnx[rn] = *reinterpret_cast<const ob_u32x2 *>(wq + (rn >> 1) * 1024 + (rn & 1) * 256);                   \
        }                                                                                                                \
        if ((STEADY) || kc + 1 < nk) { OB_G3_READ(bopA, (kc + 1) & (OB_G3_BUFS - 1), 0) }                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (OB_G3_PRIO) __builtin_amdgcn_s_setprio(1);                                                                   \
        OB_G3_MMA(bopB, 1)                                                                                               \
        if (OB_G3_PRIO) __builtin_amdgcn_s_setprio(0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if ((Q) & 1) {                                                                                                   \
            /* consume the packed words HERE (32 MFMAs after their ds_read): left pending over the step boundary */     \
            /* hipcc waits for them with lgkmcnt(0) at the top of the next step, which also drains the operand */       \
            /* reads just issued for the second half */                                                                  \
            asm volatile("" : "+v"(wnx[0]), "+v"(wnx[1]), "+v"(wnx[2]), "+v"(wnx[3]));                                   \
            _Pragma("unroll") for (int rn = 0; rn < RN; ++rn) wc[rn] = wnx[rn];                                          \
        }                                                                                                                \
    }
    ks = 0;
    for (; ks + 4 < nk; ks += 4) {              // steady state (every quad but the last): all conditions true
        OB_G3_STEP(0, true)
        OB_G3_STEP(1, true)
        OB_G3_STEP(2, true)
        OB_G3_STEP(3, true)
    }
    {                                           // the last quad: its tiles are all requested; drain
        OB_G3_STEP(0, false)
        OB_G3_STEP(1, false)
        OB_G3_STEP(2, false)
        OB_G3_STEP(3, false)
    }
#undef OB_G3_STEP
#undef OB_G3_MMA
#undef OB_G3_READ

    // Epilogue through LDS (fp16 outputs, wave's 64 rows inside N): a lane's accumulators are 4 consecutive rows
    // of one token -- stored directly that is 8 bytes per lane and 32 contiguous bytes per token row and instruction.
    // Each wave transposes its 64 x 128 tile in its own 18 KB of the (now idle: every read finished before the
    // last barrier) staging memory and writes whole 128-byte row segments, 8 token rows per instruction.
    if (!PARTIAL && n0 + wn * 64 + 64 <= N && (N & 7) == 0 && (reinterpret_cast<size_t>(u) & 15) == 0 && !(OB_G3_ABL & 32)) {
        constexpr int EP = 144;                                // bytes per token row: 128 + 16 (rows shift by 4 banks)
        char *ep = smem + (size_t)wave * 128 * EP;
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
            const int nb = n0 + wn * 64 + rn * 16 + 4 * gq;
            const ob_half4 g4 = *reinterpret_cast<const ob_half4 *>(g + nb);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                ob_half4 ov;
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = (_Float16)(ob_round_h(acc[rn][rt][i]) * (float)g4[i]);   // fp16(z) (:115), * g -> fp16 (:116)
                *reinterpret_cast<ob_half4 *>(ep + (rt * 16 + r) * EP + (rn * 16 + 4 * gq) * 2) = ov;
            }
        }
        // (same wave wrote what it reads: no barrier, the compiler orders the LDS accesses with lgkmcnt)
        _Float16 *ub = u + (int64_t)(t0 + wt * 128) * N + n0 + wn * 64;
        const int trows = T - (t0 + wt * 128);
        // Optional (ONEBIT_FLAG_TILE_STATS; `zp` is the output, N % 64 == 0 host-checked): LayerNorm partials of the rows
        // just formed -- per token and 64-row block (sum, sum of squared deviations from the block mean) of the fp16 values
        // being stored.  The 8 lanes that write a token's 128-byte segment hold its 64 values: two 8-lane DPP reductions.
        // A consumer that needs row statistics (N-sharded layers: onebit_tile_stats_combine) reads T * N / 64 pairs
        // instead of the T * N outputs again.
        float *tsp = zp;
        const int64_t ntile64 = N >> 6;
        const int tile_idx = (n0 + wn * 64) >> 6;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = j * 8 + (lane >> 3), ch = lane & 7;
            const ob_half8 v = *reinterpret_cast<const ob_half8 *>(ep + row * EP + ch * 16);
            if (row < trows) *reinterpret_cast<ob_half8 *>(ub + (int64_t)row * N + ch * 8) = v;
            if (tsp) {                                                              // (uniform)
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) sm += (float)v[i];
                sm += OB_DPP_F(sm, 0xB1, 0xF);                                      // quad_perm [1,0,3,2]
                sm += OB_DPP_F(sm, 0x4E, 0xF);                                      // quad_perm [2,3,0,1]
                sm += OB_DPP_F(sm, 0x141, 0xF);                                     // row_half_mirror: the other quad of the 8 lanes
                const float mu = sm * 0.015625f;
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) m2 = __builtin_fmaf((float)v[i] - mu, (float)v[i] - mu, m2);
                m2 += OB_DPP_F(m2, 0xB1, 0xF);
                m2 += OB_DPP_F(m2, 0x4E, 0xF);
                m2 += OB_DPP_F(m2, 0x141, 0xF);
                if (ch == 0 && row < trows)
                    *reinterpret_cast<ob_float2 *>(tsp + ((int64_t)(t0 + wt * 128 + row) * ntile64 + tile_idx) * 2) = (ob_float2){sm, m2};
            }
        }
        return;
    }
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
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)(ob_round_h(acc[rn][rt][i]) * gn[i]);
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

template <bool PARTIAL, int WT = 2>
__global__ __launch_bounds__(256 * WT, 2) void ob_gemm3_f16_kernel(
    const uint32_t *__restrict__ W, int64_t ldw_words, const _Float16 *__restrict__ a, int64_t lda,
    const _Float16 *__restrict__ g, _Float16 *__restrict__ u, float *__restrict__ zp, int T, int K, int N, int nbn)
{
    ob_gemm3_body<PARTIAL, WT>(W, ldw_words, a, lda, g, u, zp, T, K, N, nbn, ob_g3_bid());
}

// GROUPED form (round 6): up to three projections that share T, K and the row pitches -- q | k | v or gate | up of a decoder layer on the
// pre-scaled rows of THEIR OWN input_factor -- in ONE launch: tiles [tile_end[p - 1], tile_end[p]) belong to projection p.  Three separate
// launches each end in a partly filled last round of workgroups (13B, 4120 rows: 660 tiles per attention projection on 512 workgroup
// slots = 1.29 rounds, i.e. two rounds 64 % full); one launch of 1980 tiles runs 3.87 rounds.  Same tile arithmetic, bit-identical results.
struct ObG3Group {
    const uint32_t *W[3]; const _Float16 *a[3]; const _Float16 *g[3]; _Float16 *u[3];
    int N[3], nbn[3], tile_end[3];
    long long ldw_words, lda;
    int T, K;
};
template <int WT>
__global__ __launch_bounds__(256 * WT, 2) void ob_gemm3g_f16_kernel(const ObG3Group G)
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
    ob_gemm3_body<false, WT>(W, G.ldw_words, a, G.lda, g, u, nullptr, G.T, G.K, N, nbn, bid - b0);
}

// K-SLICED form (round 6): ONE projection whose 256 x 128 tiles alone leave most of the chip idle (hidden-width outputs at a few hundred
// token rows: o_proj / down_proj of a mid-size prefill or mixed step) as `ns` <= 4 K-slices in one launch -- slice j = tiles
// [j * tiles, (j + 1) * tiles) multiplies quads [q0_j, q0_j + nq_j) of K (a quad = 256 columns = 4 K steps; the longer slices first) and
// stores its fp32 sums to z[j] [T, N].  The row kernel that consumes the projection adds the slices and applies fp16(fp16(.) * g)
// (bitnet.py:115-116) to the complete sum.  ns times the workgroups, each with 1 / ns of the K loop: a round of workgroups at K = 4096 takes
// ~100 us whatever the fill, so 48 tiles x 4 slices finish in a quarter of the time of 48 tiles.
struct ObG3Slices {
    const uint32_t *W; const _Float16 *a; float *z[4];
    long long ldw_words, lda;
    int T, N, nbn, tiles, ns, quads;
};
template <int WT>
__global__ __launch_bounds__(256 * WT, 2) void ob_gemm3ks_f16_kernel(const ObG3Slices G)
{
    const int bid = ob_g3_bid();
    const int j = bid / G.tiles;
    const int qb = G.quads / G.ns, qr = G.quads - qb * G.ns;
    const int q0 = j * qb + min(j, qr), nq = qb + (j < qr ? 1 : 0);
    float *zp = j == 0 ? G.z[0] : (j == 1 ? G.z[1] : (j == 2 ? G.z[2] : G.z[3]));
    ob_gemm3_body<true, WT>(G.W + q0 * (4 * OB_G2_K / 32), G.ldw_words, G.a + q0 * (4 * OB_G2_K), G.lda, nullptr, nullptr, zp, G.T,
                            nq * (4 * OB_G2_K), G.N, G.nbn, bid - j * G.tiles);
}
#undef RN
#undef RT
