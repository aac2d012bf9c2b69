// EXPERIMENT (round 4), not part of the product.  RESULT: correct, 599 TFLOP/s (612 with the VALU phase at raised priority) against
// the product kernel's 645 -- and the ablations say why: with the softmax arithmetic removed the kernel takes 0.218 ms, with the
// MFMAs removed 0.257 ms, with both 0.463 ms: the MFMA phase of one wave and the softmax phase of its SIMD partner hardly overlap
// (HW_ID confirms waves w and w + 4 share a SIMD).  It is not the LDS traffic alone: with every LDS access removed (-DOB_FL_ABL=24,
// 25, 28) the MFMA phases take 0.147 ms, the softmax phases 0.207 ms and both 0.316 ms.  Pure MFMA and VALU streams of two waves DO
// overlap on this SIMD (tools/pipe_overlap_probe.hip: the MFMA wave keeps its rate, the VALU wave loses 15-24 %), so what serialises
// the real phases (dependent VALU chains, cross-lane swaps, the two barriers per block) is the open question.
// Without the half-period offset (-DOB_FLPP_LOCKSTEP=1: both waves of a SIMD always in the same phase) it is slower still: 572 against 606.
// Built with -DFL_PP in tools/flash_lab.hip.
//
// Causal prefill attention, head dimension 128: the arithmetic and operand layouts of ob_flash.h (read its header first), with the
// work of a CU arranged so that its matrix pipes and its VALUs are busy at the SAME time by construction instead of by chance.
//
// Workgroup = 256 queries x 8 waves (one workgroup per CU, two waves per SIMD); a wave owns 32 queries.  The waves form two groups
// (0-3 and 4-7: one wave of each group on every SIMD) that run the same loop half a period apart:
//
//     interval      2k                2k + 1              2k + 2              2k + 3
//     group 0       M(k)              V(k)                M(k + 1)            V(k + 1)
//     group 1       V(k - 1)          M(k)                V(k)                M(k + 1)
//
//   M(k) = the 64 MFMAs of a key block pair: O += V(k-1)^T P(k-1)^T, then S(k)^T = K(k) Q^T, with the fragment reads, the LDS
//          stores of block k + 1 and the global loads of block k + 2 in the shadow of the matrix pipe
//   V(k) = the softmax arithmetic of block k (mask on the diagonal, maxima, exp2, row sums, fp16 P): VALU only
// and every interval ends with one s_barrier.  While one wave of a SIMD streams MFMAs its partner does nothing but VALU work, so
// neither waits for the other's pipe (ob_flash.h's two independent workgroups per CU met in the same phase as often as not: PMC
// there shows the matrix pipe 29 % busy with the VALU and MFMA times of a wave adding up instead of overlapping).
//
// LDS: three buffers per operand (96 KB).  K(j) is read in intervals 2j, 2j + 1 and V(j) in 2j + 2, 2j + 3.  The staging rides in
// the V phases (the interval is as long as the M phase; the VALU phase has the slack): V(m) stores K(m + 2) and V(m + 1) -- 2 + 2
// sixteen-byte pieces per thread, in registers since V(m - 1) -- in intervals 2m + 1 (group 0) and 2m + 2 (group 1): after the last
// reader of the buffers' previous blocks (K(m - 1), V(m - 2): interval 2m - 1) and two barriers or more before the first reader of
// the new ones; then it starts the loads of K(m + 3) and V(m + 2).  The M phase is fragment reads and MFMAs only, and its first V
// fragments are requested at the end of the preceding V phase.
#pragma once
#include "ob_flash.h"

#ifndef OB_FLPP_PRIO
#define OB_FLPP_PRIO 0         // 1: the VALU phase runs at raised priority, 2: the MFMA phase
#endif
#ifndef OB_FLPP_LOCKSTEP
#define OB_FLPP_LOCKSTEP 0     // 1: no offset between the groups -- both waves of a SIMD are in the same phase
#endif
#define OB_FLPP_BM 256
#define OB_FLPP_NB 3
#define OB_FLPP_THREADS 512
#define OB_FLPP_LDS (2 * OB_FLPP_NB * OB_FL_BN * 128 * 2)

template <int D>
__global__ __launch_bounds__(OB_FLPP_THREADS) void ob_flash_pp_kernel(const ObFlashArgs A)
{
    constexpr int DT = D / 16, DK = D / 32, NB = OB_FLPP_NB;
    constexpr int NPC = D / 8;                          // 16-byte pieces per K / V row
    constexpr int KLD = OB_FL_BN * NPC / OB_FLPP_THREADS;   // K (and V) pieces per thread and block (2)
    constexpr int RPL = OB_FLPP_THREADS / NPC;          // key rows one pass of the 512 threads covers (32)
    constexpr int TILE = OB_FL_BN * D;                  // halves per buffer
    extern __shared__ __attribute__((aligned(16))) _Float16 ob_flpp_smem[];
    _Float16 *const Ks = ob_flpp_smem, *const Vs = ob_flpp_smem + NB * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2;
    const int lr = lane & 15, g = lane >> 4;
    const int nwg = gridDim.x, orig = blockIdx.x;       // XCD-aware numbering and causal pairing: as in ob_flash.h
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

    // staging: thread t moves piece t % NPC of rows t / NPC + RPL i (swizzles: ob_flash.h)
    ob_u32x4 kreg[KLD], vreg[KLD];
    const int srow = tid / NPC, spc = tid % NPC;
    const int kst = srow * D + 8 * (spc ^ (srow & 15)), vst = srow * D + 8 * (spc ^ ((srow & (NPC / 2 - 1)) << 1));
    auto load_k = [&](int kb) {                         // (rows past the last key read as zeros)
        const int vo = tid * 16 + kb * OB_FL_BN * D * 2;
#pragma unroll
        for (int i = 0; i < KLD; ++i) kreg[i] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, vo + i * RPL * D * 2, 0, 0));
    };
    auto load_v = [&](int kb) {
        const int vo = tid * 16 + kb * OB_FL_BN * D * 2;
#pragma unroll
        for (int i = 0; i < KLD; ++i) vreg[i] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vo + i * RPL * D * 2, 0, 0));
    };
    auto store_k = [&](int kb) {
#pragma unroll
        for (int i = 0; i < KLD; ++i) *reinterpret_cast<ob_u32x4 *>(Ks + (kb % NB) * TILE + RPL * i * D + kst) = kreg[i];
    };
    auto store_v = [&](int kb) {
#pragma unroll
        for (int i = 0; i < KLD; ++i) *reinterpret_cast<ob_u32x4 *>(Vs + (kb % NB) * TILE + RPL * i * D + vst) = vreg[i];
    };
    typedef short ob_v4s __attribute__((ext_vector_type(4)));
    typedef short ob_v8s __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) ob_v4s ob_lds_v4s;
    // Fragment addresses as LDS BYTE addresses: one long-lived register per operand -- (buffer base + lane base) ^ (d step << 6)
    // for K, ^ (d tile << 5) for V: the swizzles are XORs of whole bit fields and the buffers are 16 KB-aligned, so the tile index
    // toggles bits the rest of the address leaves zero.  (Per-tile address arrays held across the loop were spilled, and every
    // reload of one waited for vmcnt(0): for the K / V loads in flight.)
    // The lane bases themselves are recomputed in every phase from v_mbcnt (a volatile statement: otherwise they are loop
    // invariants, get spilled at this register pressure, and every reload waits for vmcnt(0) -- for the K / V loads in flight).
    typedef __attribute__((address_space(3))) char ob_lds_char;
    const uint32_t lds0 = (uint32_t)(size_t)(ob_lds_char *)ob_flpp_smem;
    auto lane_now = [&]() {
        uint32_t l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto k_base = [&](uint32_t l) { const uint32_t lr_ = l & 15, g_ = l >> 4; return lds0 + lr_ * (D * 2) + 16 * (g_ ^ lr_); };
    auto v_base = [&](uint32_t l) {
        const uint32_t lr_ = l & 15, g_ = l >> 4, vrow = 4 * g_ + (lr_ >> 2);
        return lds0 + NB * TILE * 2 + vrow * (D * 2) + 32 * (vrow & (NPC / 2 - 1)) + 16 * ((lr_ & 3) >> 1) + 8 * (lr_ & 1);
    };
    static_assert(D == 128, "the bit-field argument above is written for 16 pieces per row");

#ifdef OB_FL_TRACE
    unsigned long long *tr = nullptr;
    if (A.trace && (orig == 0 || orig == nwg / 2) && lane == 0) tr = A.trace + ((orig ? 1 : 0) * (OB_FLPP_THREADS / 64) + wave) * 66 * 8;
#endif
#ifdef OB_FL_TRACE
    if (tr) tr[65 * 8 + 4] = 0x100000000ull | __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID: SIMD_ID = bits 5:4
#endif
    for (int pass = 0; pass < 2; ++pass) {
    OB_FL_TP(0);
    const int mb = pass == 0 ? A.nmb - 1 - pj : pj;
    if (pass == 1 && mb == A.nmb - 1 - pj) break;               // odd count: the middle block has no partner
    const int m0 = mb * OB_FLPP_BM, r0 = m0 + 32 * wave;
    ob_half8 qf[2][DK];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int s = r0 + 16 * qt + lr;
        const _Float16 *qr = A.q + (((int64_t)b * S + min(s, S - 1)) * A.H + head) * D;
#pragma unroll
        for (int ds = 0; ds < DK; ++ds) qf[qt][ds] = *reinterpret_cast<const ob_half8 *>(qr + 32 * ds + 8 * g);
    }
    ob_float4 acc_o[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc_o[dt][qt] = (ob_float4){0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    const int last_q = A.past + min(m0 + OB_FLPP_BM, S) - 1;
    const int nkb = last_q / OB_FL_BN + 1;
    const int wave_first_q = A.past + r0, wave_last_q = A.past + min(r0 + 31, S - 1);
    auto is_active = [&](int kb) { return kb >= 0 && kb < nkb && kb * OB_FL_BN <= wave_last_q; };

    ob_float4 sc[4][2];         // S(k) -> V(k)
    ob_half8 pb[2][2];          // V(k) -> M(k + 1)

    // ---- M(k): O^T += V(k-1)^T . P(k-1)^T, then S(k)^T = K(k) . Q^T
    constexpr int VP = 4, KP = DK;                      // V / K fragments requested ahead
    ob_half8 vpre[VP];                                  // V(k) -> M(k + 1): the first V fragments of block k
    auto read_v = [&](uint32_t vbuf, int j) {            // fragment j = (key step j / DT, d tile j % DT): two transpose reads
        const uint32_t a = (vbuf ^ ((j % DT) << 5)) + 32 * (j / DT) * D * 2;
        const ob_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(ob_lds_char *)(size_t)a);
        const ob_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(ob_lds_char *)(size_t)(a + 16 * D * 2));
        const ob_v8s a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(ob_half8, a8);
    };
    auto m_phase = [&](const int k) {
        const bool do_pv = !(OB_FL_ABL & 4) && is_active(k - 1), do_s = !(OB_FL_ABL & 4) && is_active(k);
        const uint32_t lane_m = lane_now();
        const uint32_t kbuf = k_base(lane_m) + (k % NB) * (TILE * 2);
        ob_half8 kf[4 * DK];
        typedef __attribute__((address_space(3))) const ob_half8 ob_lds_half8;
        auto read_k = [&](int i) { if (!(OB_FL_ABL & 8)) kf[i] = *(ob_lds_half8 *)(ob_lds_char *)(size_t)((kbuf ^ ((i % DK) << 6)) + 16 * (i / DK) * D * 2); else kf[i] = qf[0][i % DK]; };
        if (do_s) {                                     // in flight underneath the output MFMAs
#pragma unroll
            for (int i = 0; i < KP; ++i) read_k(i);
        }
        if (do_pv) {
            const uint32_t Vb = v_base(lane_m) + ((k - 1) % NB) * (TILE * 2);
            ob_half8 vf[2 * DT];
#pragma unroll
            for (int j = 0; j < VP; ++j) vf[j] = vpre[j];
#pragma unroll
            for (int j = 0; j < 2 * DT; ++j) {
                if (j + VP < 2 * DT) vf[j + VP] = (OB_FL_ABL & 8) ? qf[1][j % DK] : read_v(Vb, j + VP);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) acc_o[j % DT][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[j], pb[qt][j / DT], acc_o[j % DT][qt], 0, 0, 0);
                if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (do_s) {
#pragma unroll
            for (int i = 0; i < 4 * DK; ++i) {
                if (i + KP < 4 * DK) read_k(i + KP);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    sc[i / DK][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i], qf[qt][i % DK], i % DK == 0 ? (ob_float4){0.f, 0.f, 0.f, 0.f} : sc[i / DK][qt], 0, 0, 0);
                if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // ---- V(k): online softmax of block k (ob_flash.h: the running maximum moves only when a tile exceeds it by more than 2^THR)
    auto v_phase = [&](const int k) {
        if (!(OB_FL_ABL & 16)) {
        store_k(k + 2);                                 // (blocks past the last one: zeros, never read)
        store_v(k + 1);
        load_k(k + 3);
        load_v(k + 2);
        }
        if ((OB_FL_ABL & 1) || !is_active(k)) return;
        const int k0 = k * OB_FL_BN;
        const bool diag = k0 + OB_FL_BN - 1 > wave_first_q || k0 + OB_FL_BN > L;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (diag) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kidx = k0 + 16 * kt + 4 * g + e;
                        if (kidx > A.past + r0 + 16 * qt + lr || kidx >= L) sc[kt][qt][e] = -INFINITY;
                    }
            }
            float mx = sc[0][qt][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sc[kt][qt][e]);
            mx = ob_fl_col_max(mx);
            if (__builtin_amdgcn_ballot_w64((mx - m_run[qt]) * A.scale_log2e > OB_FL_DEFER_THR) != 0) {
                const float m_new = fmaxf(m_run[qt], mx);
                const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m_run[qt] - m_new) * A.scale_log2e);
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc_o[dt][qt] *= alpha;
                m_run[qt] = m_new;
            }
            const float nm = m_run[qt] == -INFINITY ? 0.f : -m_run[qt] * A.scale_log2e;
            float ls = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][qt][e], A.scale_log2e, nm));
                    sc[kt][qt][e] = p;
                    ls += p;
                }
            l_run[qt] += ls;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pb[qt][ks][e] = (_Float16)sc[2 * ks][qt][e];
                    pb[qt][ks][4 + e] = (_Float16)sc[2 * ks + 1][qt][e];
                }
        }
        const uint32_t Vn = v_base(lane_now()) + (k % NB) * (TILE * 2);
#pragma unroll
        for (int j = 0; j < VP; ++j) vpre[j] = (OB_FL_ABL & 24) ? qf[0][j % DK] : read_v(Vn, j);      // (V(k): in LDS since the V(k - 1) phases)
    };

    load_k(0); load_v(0);
    store_k(0); store_v(0);
    load_k(1);
    store_k(1);
    load_k(2); load_v(1);
    __syncthreads();
    OB_FL_TP(1);
    if (!OB_FLPP_LOCKSTEP && grp == 1) __syncthreads(); // group 1 runs half a period behind
    for (int kb = 0; kb <= nkb; ++kb) {
        OB_FL_T(0);
#if OB_FLPP_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        m_phase(kb);
#if OB_FLPP_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        OB_FL_T(1);
        __syncthreads();
        OB_FL_T(2);
#if OB_FLPP_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
        v_phase(kb);
#if OB_FLPP_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        OB_FL_T(3);
        __syncthreads();
        OB_FL_T(4);
    }
    if (!OB_FLPP_LOCKSTEP && grp == 0) __syncthreads();
    OB_FL_TP(2);

    // ---- normalise and write: lane holds d = 16 dt + 4 g .. + 3 of query lr of each tile
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
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
