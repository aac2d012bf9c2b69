// EXPERIMENT (round 4), not part of the product: causal prefill attention with one wave per SIMD -- workgroup = 256 queries x 4 waves, a
// wave owns 64 queries and the whole 512-entry register file, three LDS buffers per operand, MFMAs as asm statements so that the
// register FILE of every operand is chosen here (with the builtin, hipcc kept the output accumulators in AGPRs and moved 128
// registers per key block to the VGPRs for the rare rescale), and the softmax of one half of the queries hand-interleaved with the
// other half's MFMAs.  Correct at every shape tools/flash_lab.hip checks, but 565 TFLOP/s against the product kernel's 654 at
// 8 x 2048 x 32 x 128 (profiles/r04_attention_probe.txt): with one wave per SIMD nothing covers the stalls that remain (PMC: VALU
// issue 36 % of the wave cycles, MFMA busy 27 %, waiting 30 %; 480 VALU instructions per 128 MFMAs; L2 hit rate of the K / V stream
// 56 %), and the diagonal blocks (4 per 256 queries, with idle waves) and the per-pass prologue / epilogue cost 40 % of the time.
// Kept for the record of what was measured and for the two findings that generalise:
//   * an MFMA inside an asm statement gets none of the wait states a VALU result needs before a matrix instruction may read it:
//     with the producers sunk next to the MFMA the kernel returned inf; ob_fl_pin() materialises such values where they are made
//   * a burst of 8 buffer loads in a VALU-bound phase costs ~270 cycles per load
// Build: hipcc ... -DFL_K64 tools/flash_lab.hip (includes this file after ob_flash.h).
#pragma once
#include <utility>
#include "ob_flash.h"

// ---------------------------------------------------------------------------------------------------------------------
// D = 128, one wave per SIMD: workgroup = 256 queries x 4 waves, a wave owns 64 queries (four 16-query tiles) and the whole
// 512-entry register file.  Same arithmetic and operand layouts as above; what changes is the ratio of everything that is not
// an MFMA to the MFMAs: a K fragment read feeds 4 MFMAs instead of 2, a V fragment (two transpose reads) 4 instead of 2, a
// staged K / V piece is shared by twice the queries -- per 128 MFMAs of a key block a wave issues 48 LDS reads, 8 + 8 staging
// instructions and ~230 VALU, which fits in the shadow of the matrix pipe (an MFMA occupies it for 16 cycles, the wave issues
// in 4).  Three LDS buffers per operand: block kb + 2 is written while kb is computed, so the fragments of block kb + 1 may be
// requested BEFORE the barrier that ends block kb (its first MFMAs find their operands in registers).
// MFMAs with the register FILE of every operand chosen here, not by the allocator: with the builtin, hipcc put the score
// accumulators into AGPRs and the output accumulators where the (rare) rescale wanted them, and moved 128 registers between the
// two files per key block.  Here everything the VALU touches (output and score accumulators, P) is in VGPRs and everything that
// only travels LDS -> MFMA (Q, K, V fragments) in AGPRs.  The allocator does not see an MFMA in these statements, so it inserts none of the wait states an
// MFMA result needs before a VALU instruction may read it: ob_fl_mfma_settle() follows every group whose results are read next.
__device__ __forceinline__ void ob_fl_mfma_s0(ob_float4 &c, const ob_half8 &a, const ob_half8 &b)     // scores: D in VGPRs, A (K) and B (Q) in AGPRs, C = 0
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "a"(a), "a"(b));
}
__device__ __forceinline__ void ob_fl_mfma_s(ob_float4 &c, const ob_half8 &a, const ob_half8 &b)
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "a"(b));
}
__device__ __forceinline__ void ob_fl_mfma_o(ob_float4 &c, const ob_half8 &a, const ob_half8 &b)      // output: C / D in VGPRs (the rescale is VALU work), A (V) in AGPRs
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
}
template <class F, int... I>
__device__ __forceinline__ void ob_fl_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>()), ...); }
template <int N, class F>
__device__ __forceinline__ void ob_fl_static_for(F &&f) { ob_fl_static_for_impl(f, std::make_integer_sequence<int, N>()); }
// A value the VALU wrote and an MFMA statement will read: materialised HERE (the allocator is free to sink its producer down to the
// first use it can see, and an MFMA inside an asm statement gets none of the wait states a VALU result needs before a matrix
// instruction may read it -- tests/test_flash_isa.py checks the generated code for such pairs)
template <class T>
__device__ __forceinline__ void ob_fl_pin(T &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void ob_fl_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 7"); }

#define OB_FL64_BM 256
#define OB_FL64_NB 3
#define OB_FL64_LDS (2 * OB_FL64_NB * OB_FL_BN * 128 * 2)
#ifndef OB_FL64_PIPE
#define OB_FL64_PIPE 1      // 0: the plain (phase after phase) body for every block
#endif
#ifndef OB_FL64_KPRE
#define OB_FL64_KPRE 1      // K fragment chunks (key tiles) requested ahead
#endif
#ifndef OB_FL64_VPRE
#define OB_FL64_VPRE 1      // V fragment chunks requested ahead
#endif

template <int D>
__global__ __launch_bounds__(256, 1) void ob_flash_fwd64_kernel(const ObFlashArgs A)
{
    constexpr int DT = D / 16, DK = D / 32, QT = 4;
    constexpr int NPC = D / 8, KLD = OB_FL_BN * NPC / 256, RPL = 256 / NPC, NB = OB_FL64_NB;
    constexpr int TILE = OB_FL_BN * D;          // halves per K (or V) buffer
    extern __shared__ __attribute__((aligned(16))) _Float16 ob_fl_smem[];
    _Float16 *const Ks = ob_fl_smem, *const Vs = ob_fl_smem + NB * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, rem = nwg & 7;
    const int bid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (orig >> 3);
    const int npair = (A.nmb + 1) >> 1;
    const int pj = bid % npair;
    const int bh = bid / npair;
    const int head = bh % A.H, b = bh / A.H;
    const int kvh = head / (A.H / A.Hkv);
    const int S = A.S, L = A.past + S;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.k + ((int64_t)b * A.Hkv + kvh) * A.max_len * D), 0, L * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.v + ((int64_t)b * A.Hkv + kvh) * A.max_len * D), 0, L * D * 2, 0x00020000);

    // staging (thread t moves piece t % NPC of rows t / NPC + RPL i) and operand addresses: as in the kernel above
    ob_u32x4 kreg[KLD], vreg[KLD];
    const int srow = tid / NPC, spc = tid % NPC;
    const int kst = srow * D + 8 * (spc ^ (srow & 15)), vst = srow * D + 8 * (spc ^ ((srow & (NPC / 2 - 1)) << 1));
    auto load_block = [&](int kb) {
        const int vo = tid * 16 + kb * OB_FL_BN * D * 2;
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            kreg[i] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, vo + i * RPL * D * 2, 0, 0));
            vreg[i] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vo + i * RPL * D * 2, 0, 0));
        }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            *reinterpret_cast<ob_u32x4 *>(Ks + buf * TILE + RPL * i * D + kst) = kreg[i];
            *reinterpret_cast<ob_u32x4 *>(Vs + buf * TILE + RPL * i * D + vst) = vreg[i];
        }
    };
    typedef short ob_v4s __attribute__((ext_vector_type(4)));
    typedef short ob_v8s __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) ob_v4s ob_lds_v4s;
    int koff[DK], voff[DT];
#pragma unroll
    for (int ds = 0; ds < DK; ++ds) koff[ds] = lr * D + 8 * ((4 * ds + g) ^ lr);
    const int vrow = 4 * g + (lr >> 2), vswz_r = vrow & (NPC / 2 - 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) voff[dt] = vrow * D + 8 * (2 * (dt ^ vswz_r) + ((lr & 3) >> 1)) + 4 * (lr & 1);
    auto read_k = [&](const _Float16 *Kb, int i) {          // fragment i = (key tile i / DK, d step i % DK)
        return *reinterpret_cast<const ob_half8 *>(Kb + 16 * (i / DK) * D + koff[i % DK]);
    };
    auto read_v = [&](const _Float16 *Vb, int j) {          // fragment j = (key step j / DT, d tile j % DT): two transpose reads
        const ob_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(Vb + 32 * (j / DT) * D + voff[j % DT]));
        const ob_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(Vb + (32 * (j / DT) + 16) * D + voff[j % DT]));
        const ob_v8s a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(ob_half8, a8);
    };

#ifdef OB_FL_TRACE
    unsigned long long *tr = nullptr;
    if (A.trace && (orig == 0 || orig == nwg / 2) && lane == 0) tr = A.trace + ((orig ? 1 : 0) * (OB_FL_THREADS / 64) + wave) * 66 * 8;
#endif
    for (int pass = 0; pass < 2; ++pass) {
    OB_FL_TP(0);
    const int mb = pass == 0 ? A.nmb - 1 - pj : pj;
    if (pass == 1 && mb == A.nmb - 1 - pj) break;
    const int m0 = mb * OB_FL64_BM, r0 = m0 + 64 * wave;
    ob_half8 qf[QT][DK];
    int qpos[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int s = r0 + 16 * qt + lr;
        qpos[qt] = A.past + s;
        const _Float16 *qr = A.q + (((int64_t)b * S + min(s, S - 1)) * A.H + head) * D;
#pragma unroll
        for (int ds = 0; ds < DK; ++ds) qf[qt][ds] = *reinterpret_cast<const ob_half8 *>(qr + 32 * ds + 8 * g);
    }
    ob_float4 acc_o[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { acc_o[dt][qt] = (ob_float4){0.f, 0.f, 0.f, 0.f}; ob_fl_pin(acc_o[dt][qt]); }
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    const int last_q = A.past + min(m0 + OB_FL64_BM, S) - 1;
    const int nkb = last_q / OB_FL_BN + 1;
    const int wave_last_q = A.past + min(r0 + 63, S - 1);
    const int nfull = min((A.past + m0 + 1) / OB_FL_BN, nkb);   // blocks entirely below every wave's diagonal

    constexpr int KPRE = 4 * DK;                // the K fragments of a block are requested during the previous block's last MFMAs
    ob_half8 kpre[KPRE];
    auto block = [&](const int kb, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        OB_FL_T(0);
        const int k0 = kb * OB_FL_BN;
        const int buf = kb % NB, buf1 = (kb + 1) % NB, buf2 = (kb + 2) % NB;
        const bool active = !TAIL || k0 <= wave_last_q;
        const bool diag = TAIL && (k0 + OB_FL_BN - 1 > A.past + r0 || k0 + OB_FL_BN > L);
        const _Float16 *Kb = Ks + buf * TILE, *Vb = Vs + buf * TILE;
        // invariant at entry (both bodies): blocks <= kb + 1 are in LDS, the staging registers are free.  Block kb + 2 leaves memory
        // now and goes to its buffer (last read in block kb - 1) before the output MFMAs.
        load_block(kb + 2);
        if (active) {
            // K fragments of the whole block (AGPRs): on their way since the previous iteration
            ob_half8 kf[4 * DK];
#pragma unroll
            for (int i = 0; i < KPRE; ++i) kf[i] = kpre[i];
            constexpr int VC = 4, NVC = 2 * DT / VC;        // V fragments in chunks of 4 (= 16 MFMAs)
            ob_half8 vf[2 * DT];
            ob_half8 pb[QT][2];
            // The 64 queries are taken in two halves of two tiles: 32 score registers instead of 64 (the output accumulators
            // alone are 128 of the 256 VGPRs)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // ---- S^T = K . Q^T: 4 key tiles x 2 query tiles
                ob_float4 sc[4][2];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int ds = 0; ds < DK; ++ds)
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2) {
                            if (ds == 0) ob_fl_mfma_s0(sc[c][q2], kf[c * DK + ds], qf[2 * h + q2][ds]);
                            else ob_fl_mfma_s(sc[c][q2], kf[c * DK + ds], qf[2 * h + q2][ds]);
                        }
                if (h == 0) OB_FL_T(1); else OB_FL_T(3);
                if (h == 1) {
#pragma unroll
                    for (int j = 0; j < OB_FL64_VPRE * VC; ++j) vf[j] = read_v(Vb, j);      // in flight underneath the softmax arithmetic
                }
                ob_fl_mfma_settle();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int qt = 2 * h + q2;
                    if (TAIL && diag) {
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int kidx = k0 + 16 * kt + 4 * g + e;
                                if (kidx > qpos[qt] || kidx >= L) sc[kt][q2][e] = -INFINITY;
                            }
                    }
                    float mx = sc[0][q2][0];
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sc[kt][q2][e]);
                    mx = ob_fl_col_max(mx);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64((mx - m_run[qt]) * A.scale_log2e > OB_FL_DEFER_THR) != 0, 0)) {      // (see the kernel above)
                        const float m_new = fmaxf(m_run[qt], mx);
                        const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m_run[qt] - m_new) * A.scale_log2e);
                        l_run[qt] *= alpha;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) { acc_o[dt][qt] *= alpha; ob_fl_pin(acc_o[dt][qt]); }
                        m_run[qt] = m_new;
                    }
                    const float nm = m_run[qt] == -INFINITY ? 0.f : -m_run[qt] * A.scale_log2e;
                    float ls = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][q2][e], A.scale_log2e, nm));
                            sc[kt][q2][e] = p;
                            ls += p;
                        }
                    l_run[qt] += ls;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pb[qt][ks][e] = (_Float16)sc[2 * ks][q2][e];
                            pb[qt][ks][4 + e] = (_Float16)sc[2 * ks + 1][q2][e];
                        }
                    ob_fl_pin(pb[qt][0]);
                    ob_fl_pin(pb[qt][1]);
                }
                asm volatile("s_nop 7");                // (P written by the VALU, read by the MFMAs of the output product)
                __builtin_amdgcn_sched_barrier(0);
                if (h == 0) OB_FL_T(2); else OB_FL_T(4);
            }
            store_block(buf2);
            // ---- O^T += V^T . P^T, the other V fragments and the next block's K fragments requested along the way
#pragma unroll
            for (int c = 0; c < NVC; ++c) {
                if (c + OB_FL64_VPRE < NVC) {
#pragma unroll
                    for (int j = 0; j < VC; ++j) vf[(c + OB_FL64_VPRE) * VC + j] = read_v(Vb, (c + OB_FL64_VPRE) * VC + j);
                }
#pragma unroll
                for (int i = 0; i < DK; ++i) kpre[c * DK + i] = read_k(Ks + buf1 * TILE, c * DK + i);
#pragma unroll
                for (int j = c * VC; j < (c + 1) * VC; ++j)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) ob_fl_mfma_o(acc_o[j % DT][qt], vf[j], pb[qt][j / DT]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            store_block(buf2);
#pragma unroll
            for (int i = 0; i < KPRE; ++i) kpre[i] = read_k(Ks + buf1 * TILE, i);      // block kb + 1 has been in LDS since iteration kb - 1
        }
        OB_FL_T(5);
        __syncthreads();
        OB_FL_T(6);
    };
    // Blocks below every diagonal, software-pipelined by hand: the MFMAs are asm statements in program order, the softmax of one
    // half of the queries is cut into 32 steps of ~5 VALU instructions and one step follows each MFMA of the OTHER half's
    // matrix product (an MFMA occupies the matrix pipe for 16 cycles and the wave's issue slot for 4):
    //   A  scores(h0)            + V fragment reads       B  scores(h1) + softmax(h0)
    //   C  output(h0) + softmax(h1)                       D  output(h1)            + K fragment reads of the next block
    // Same arithmetic in the same order as the plain body (which the diagonal blocks keep using).
    auto block_main = [&](const int kb) {
        OB_FL_T(0);
        const int buf = kb % NB, buf1 = (kb + 1) % NB, buf2 = (kb + 2) % NB;
        const _Float16 *vp[DT], *kp[DK];       // this lane's fragment addresses in V block kb / K block kb + 1: constants away from every read
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vp[dt] = Vs + buf * TILE + voff[dt];
#pragma unroll
        for (int ds = 0; ds < DK; ++ds) kp[ds] = Ks + buf1 * TILE + koff[ds];
        auto read_vp = [&](int j) {
            const ob_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(vp[j % DT] + 32 * (j / DT) * D));
            const ob_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(vp[j % DT] + (32 * (j / DT) + 16) * D));
            const ob_v8s a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(ob_half8, a8);
        };
        ob_half8 (&kf)[4 * DK] = kpre;      // requested during the previous block's phase D
        ob_half8 vf[2 * DT], pb[QT][2];
        // block kb + 2 leaves memory in 2 KLD pieces spread over phases A - C (a burst of buffer loads stalls the wave on the
        // texture path, and with it the matrix pipe), and goes to its LDS buffer during phase D
        auto load_piece = [&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            const int vo = tid * 16 + (kb + 2) * OB_FL_BN * D * 2;
            if constexpr (i < KLD) kreg[i] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, vo + i * RPL * D * 2, 0, 0));
            else vreg[i - KLD] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vo + (i - KLD) * RPL * D * 2, 0, 0));
        };
        ob_float4 sc0[4][2], sc1[4][2];
        float mx[2], nm[2], ls[2];
        bool need[2];
        // one softmax step: n = 16 q2 + s for query tile qb + q2 of the half whose scores are sc
        auto sm_step = [&](auto n_c, ob_float4 (&sc)[4][2], auto qb_c) {
            constexpr int n = decltype(n_c)::value, q2 = n / 16, st = n % 16, qt = decltype(qb_c)::value + q2;
            if constexpr (st < 4) {
                float t = sc[st][q2][0];
#pragma unroll
                for (int e = 1; e < 4; ++e) t = fmaxf(t, sc[st][q2][e]);
                mx[q2] = st == 0 ? t : fmaxf(mx[q2], t);
            } else if constexpr (st == 4) {
                mx[q2] = ob_fl_col_max(mx[q2]);
                need[q2] = __builtin_amdgcn_ballot_w64((mx[q2] - m_run[qt]) * A.scale_log2e > OB_FL_DEFER_THR) != 0;
            } else if constexpr (st == 5) {
                if (__builtin_expect(need[q2], 0)) {
                    const float m_new = fmaxf(m_run[qt], mx[q2]);
                    const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m_run[qt] - m_new) * A.scale_log2e);
                    l_run[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { acc_o[dt][qt] *= alpha; ob_fl_pin(acc_o[dt][qt]); }
                    m_run[qt] = m_new;
                }
                nm[q2] = m_run[qt] == -INFINITY ? 0.f : -m_run[qt] * A.scale_log2e;
                ls[q2] = 0.f;
            } else if constexpr (st < 14) {
                constexpr int i = st - 6, kt = i / 2, e0 = (i & 1) * 2;
#pragma unroll
                for (int e = e0; e < e0 + 2; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][q2][e], A.scale_log2e, nm[q2]));
                    sc[kt][q2][e] = pv;
                    ls[q2] += pv;
                }
            } else {
                constexpr int ks = st - 14;
                if constexpr (ks == 0) l_run[qt] += ls[q2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pb[qt][ks][e] = (_Float16)sc[2 * ks][q2][e];
                    pb[qt][ks][4 + e] = (_Float16)sc[2 * ks + 1][q2][e];
                }
                ob_fl_pin(pb[qt][ks]);
            }
        };
        // A
        ob_fl_static_for<8 * DK>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value, c = n / (2 * DK), ds = (n / 2) % DK, q2 = n & 1;
            if constexpr (ds == 0) ob_fl_mfma_s0(sc0[c][q2], kf[c * DK + ds], qf[q2][ds]);
            else ob_fl_mfma_s(sc0[c][q2], kf[c * DK + ds], qf[q2][ds]);
            if constexpr ((n & 1) == 1 && (n >> 1) < 2 * DT) vf[n >> 1] = read_vp(n >> 1);
            if constexpr ((n & 7) == 3 && (n >> 3) < 2 * KLD) load_piece(std::integral_constant<int, (n >> 3)>());         // pieces 0 - 3
            if constexpr ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        });
        ob_fl_mfma_settle();
        OB_FL_T(1);
        // B
        ob_fl_static_for<8 * DK>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value, c = n / (2 * DK), ds = (n / 2) % DK, q2 = n & 1;
            if constexpr (ds == 0) ob_fl_mfma_s0(sc1[c][q2], kf[c * DK + ds], qf[2 + q2][ds]);
            else ob_fl_mfma_s(sc1[c][q2], kf[c * DK + ds], qf[2 + q2][ds]);
            sm_step(n_c, sc0, std::integral_constant<int, 0>());
            if constexpr ((n & 7) == 3 && 4 + (n >> 3) < 2 * KLD) load_piece(std::integral_constant<int, 4 + (n >> 3)>());   // pieces 4 - 7
            __builtin_amdgcn_sched_barrier(0);
        });
        ob_fl_mfma_settle();
        OB_FL_T(2);
        __syncthreads();            // the one barrier of the block, where the LDS is quiet: orders phase D's stores (previous block) and reads (this block)
        // C
        ob_fl_static_for<4 * DT>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value, j = n >> 1, q2 = n & 1;
            ob_fl_mfma_o(acc_o[j % DT][q2], vf[j], pb[q2][j / DT]);
            sm_step(n_c, sc1, std::integral_constant<int, 2>());
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 7");
        OB_FL_T(3);
        // D: behind the 32 MFMAs, block kb + 2 goes to its buffer (last read in block kb - 1) and the K fragments of block kb + 1
        // (in LDS since the previous block) are requested
        ob_fl_static_for<4 * DT>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value, j = n >> 1, q2 = n & 1;
            ob_fl_mfma_o(acc_o[j % DT][2 + q2], vf[j], pb[2 + q2][j / DT]);
            if constexpr ((n & 1) == 0 && (n >> 1) < 2 * KLD) {
                constexpr int i = n >> 1;
                if constexpr (i < KLD) *reinterpret_cast<ob_u32x4 *>(Ks + buf2 * TILE + RPL * i * D + kst) = kreg[i];
                else *reinterpret_cast<ob_u32x4 *>(Vs + buf2 * TILE + RPL * (i - KLD) * D + vst) = vreg[i - KLD];
            }
            if constexpr ((n & 1) == 1 && (n >> 1) < KPRE) kpre[n >> 1] = *reinterpret_cast<const ob_half8 *>(kp[(n >> 1) % DK] + 16 * ((n >> 1) / DK) * D);
            if constexpr ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        });
        OB_FL_T(5);
    };
    load_block(0);
    store_block(0);
    load_block(1);
    store_block(1 % NB);
    __syncthreads();
    OB_FL_TP(1);
#pragma unroll
    for (int i = 0; i < KPRE; ++i) kpre[i] = read_k(Ks, i);
    int kb = 0;
    for (; kb < nfull; ++kb) {
        if (OB_FL64_PIPE) block_main(kb);
        else block(kb, std::false_type());
    }
    if (OB_FL64_PIPE && nfull > 0) {    // (the pipelined body ends without a barrier: its last stores and fragment reads are still unordered)
        __syncthreads();
    }
    for (; kb < nkb; ++kb) block(kb, std::true_type());
    ob_fl_mfma_settle();
    OB_FL_TP(2);

    // ---- normalise and write
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int s = r0 + 16 * qt + lr;
        const float l = ob_fl_col_sum(l_run[qt]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (s >= S) continue;
        _Float16 *orow = A.o + (((int64_t)b * S + s) * A.H + head) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            ob_half4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (_Float16)(acc_o[dt][qt][e] * inv);
            if (A.h_next) ov = ov * *reinterpret_cast<const ob_half4 *>(A.h_next + head * D + 16 * dt + 4 * g);
            *reinterpret_cast<ob_half4 *>(orow + 16 * dt + 4 * g) = ov;
        }
    }
    OB_FL_TP(3);
    }
}
