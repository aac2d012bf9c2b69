// Causal prefill attention for gfx950 (modeling_bitllama.py:546-563: scores = q k^T / sqrt(D) + causal mask, softmax in
// fp32, probabilities . v), flash style: no [S, S] score tensor in HBM, fp32 online softmax, MFMA 16x16x32 f16.
// Replaces the vendor's fused attention (a Triton kernel behind torch SDPA) on the fused prefill route.
//
// Work decomposition: workgroup = (128 queries of one head of one sequence) x 4 waves, a wave owns 32 queries (two
// 16-query tiles) and sweeps the keys in blocks of 64 through LDS.  Everything is computed TRANSPOSED so that the
// probabilities never leave registers between the two matrix products:
//   S^T[key][query] = K . Q^T      A = K rows (lane: key = lane & 15, 8 consecutive d: one 16-byte LDS read of the
//                                  row-major K tile), B = Q^T (lane: query = lane & 15, 8 consecutive d: loaded once)
//                                  -> a lane holds, per 16-key tile, 4 consecutive keys of ONE query
//   O^T[d][query]  = V^T . P^T     B = P^T: the lane's 4 + 4 keys of two adjacent key tiles ARE its 8 k-elements (the
//                                  k order of a product is free as long as both operands agree), A = V^T rows
//                                  (lane: d = lane & 15, the same 8 keys) -- V is transposed once on its way into LDS
//                                  (4 keys x 8 d per thread, a register transpose, 8-byte LDS stores in that key order)
// so the softmax statistics of a query live in ONE lane column (lane & 15) across the 4 lane groups: row maxima / sums
// are in-lane reductions plus two row swaps (v_permlane16_swap / v_permlane32_swap), and the rescale factor of the
// running output is a per-lane scalar.  Output: 4 consecutive d of one query per lane and tile, written token-major
// [B, S, H, D] = the rows o_proj consumes, optionally already multiplied by o_proj's input_factor (bitnet.py:113) so
// that the projection runs with ONEBIT_FLAG_PRESCALED.
// Causality: key blocks above the diagonal are never loaded; diagonal blocks are masked per element; query blocks are
// issued heaviest first.
#pragma once
#include "ob_common.h"

struct ObFlashArgs {
    const _Float16 *q;        // [B, S, H, D] token-major (onebit_rows_qkv_rope with ONEBIT_FLAG_Q_TOKEN_MAJOR)
    const _Float16 *k, *v;    // cache rows [B][Hkv][max_len][D]; keys 0 .. past + S - 1 are valid
    _Float16 *o;              // [B, S, H, D]
    const _Float16 *h_next;   // optional [H * D]: o <- fp16(o * h_next)
    int S, H, Hkv, max_len, past;
    float scale_log2e;        // log2(e) / sqrt(D)
    int nmb;                  // query blocks per (batch, head)
};

#define OB_FL_BM 128
#define OB_FL_BN 64

__device__ __forceinline__ float ob_fl_col_max(float v)      // max over the 4 lanes that share lane & 15
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float ob_fl_col_sum(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int D>
__global__ __launch_bounds__(256, 2) void ob_flash_fwd_kernel(const ObFlashArgs A)
{
    constexpr int DT = D / 16, DK = D / 32;
    constexpr int KP = D + 8;                   // halves per K row in LDS (16-byte pad)
    constexpr int VP = OB_FL_BN + 8;            // halves per V^T row
    constexpr int NPC = D / 8;                  // 16-byte pieces per K / V row
    constexpr int KLD = OB_FL_BN * NPC / 256;   // K pieces per thread and block (4 at D = 128)
    __shared__ __attribute__((aligned(16))) _Float16 Ks[OB_FL_BN][KP];
    __shared__ __attribute__((aligned(16))) _Float16 Vt[D][VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    // XCD-aware numbering: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so
    // the logical block ids are renumbered to give every XCD one CONTIGUOUS range -- the 16 query blocks of a (sequence,
    // head) then run on one XCD, close in time, and its K / V rows (1 MB at S = 2048) are fetched from HBM once instead
    // of once per XCD and query block (measured: 1.02 -> see DESIGN.md; dealt plainly, 8 x 2048 x 32 heads re-read
    // 2.2 GB of K / V through thrashing L2s).  Within a head: heaviest query blocks first (they sweep the most keys).
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, rem = nwg & 7;
    const int bid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (orig >> 3);
    const int mb = A.nmb - 1 - (bid % A.nmb);
    const int bh = bid / A.nmb;
    const int head = bh % A.H, b = bh / A.H;
    const int kvh = head / (A.H / A.Hkv);
    const int m0 = mb * OB_FL_BM;
    const int S = A.S, L = A.past + S;
    const _Float16 *kb_ = A.k + ((int64_t)b * A.Hkv + kvh) * A.max_len * D;
    const _Float16 *vb_ = A.v + ((int64_t)b * A.Hkv + kvh) * A.max_len * D;

    // Q^T fragments of the wave's two query tiles: lane (query = lr, d = 32 ds + 8 g .. + 7)
    ob_half8 qf[2][DK];
    int qpos[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int s = m0 + 32 * wave + 16 * qt + lr;
        qpos[qt] = A.past + s;                                  // absolute position of this lane's query
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

    // key blocks this workgroup needs: keys 0 .. past + min(m0 + 128, S) - 1
    const int last_q = A.past + min(m0 + OB_FL_BM, S) - 1;
    const int nkb = last_q / OB_FL_BN + 1;
    const int wave_last_q = A.past + min(m0 + 32 * wave + 31, S - 1);      // beyond it every key is masked for this wave

    // staging: K piece (key = p / NPC, d = 8 (p % NPC)); V quad (keys 4 kq .. + 3, d = 8 dg .. + 7)
    const int vkq = tid / NPC, vdg = tid % NPC;
    const bool vact = vkq < OB_FL_BN / 4;
    ob_u32x4 kreg[KLD], vreg[4];
    auto load_block = [&](int kb) {
        const int k0 = kb * OB_FL_BN;
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int p = tid + 256 * i, key = p / NPC, pc = p % NPC;
            kreg[i] = *reinterpret_cast<const ob_u32x4 *>(kb_ + (int64_t)min(k0 + key, L - 1) * D + 8 * pc);
        }
        if (vact) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                vreg[r] = *reinterpret_cast<const ob_u32x4 *>(vb_ + (int64_t)min(k0 + 4 * vkq + r, L - 1) * D + 8 * vdg);
        }
    };
    auto store_block = [&]() {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int p = tid + 256 * i, key = p / NPC, pc = p % NPC;
            *reinterpret_cast<ob_u32x4 *>(&Ks[key][8 * pc]) = kreg[i];
        }
        if (vact) {
            // 4 keys x 8 d -> 8 rows (d) of 4 keys: position of key k inside its chunk of 32 = 8 ((k % 16) / 4) + 4 ((k % 32) / 16) + k % 4.
            // The 16 lanes of a key quad write rows 8 apart (36 dwords each: the same banks); the 8-half column groups of
            // a row are therefore XOR-swizzled with (row / 8) % 8 -- 2-way instead of 16-way conflicts, reads use the same map
            const int kq = 4 * vkq;
            const int pos = (kq / 32) * 32 + 8 * ((kq % 16) / 4) + 4 * ((kq % 32) / 16);
            const int swz = (vdg & 7) << 3;                    // rows 8 vdg .. + 7: (row >> 3) & 7 = vdg & 7
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                // dwords e2 of the four key rows hold d = 8 dg + 2 e2, + 1 of keys 0..3
                const uint32_t a0 = vreg[0][e2], a1 = vreg[1][e2], a2 = vreg[2][e2], a3 = vreg[3][e2];
                const uint32_t lo01 = __builtin_amdgcn_perm(a1, a0, 0x05040100u), lo23 = __builtin_amdgcn_perm(a3, a2, 0x05040100u);
                const uint32_t hi01 = __builtin_amdgcn_perm(a1, a0, 0x07060302u), hi23 = __builtin_amdgcn_perm(a3, a2, 0x07060302u);
                *reinterpret_cast<ob_u32x2 *>(&Vt[8 * vdg + 2 * e2][pos ^ swz]) = (ob_u32x2){lo01, lo23};
                *reinterpret_cast<ob_u32x2 *>(&Vt[8 * vdg + 2 * e2 + 1][pos ^ swz]) = (ob_u32x2){hi01, hi23};
            }
        }
    };

    load_block(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();                        // the previous block's tiles have been consumed
        store_block();
        if (kb + 1 < nkb) load_block(kb + 1);   // in flight underneath this block's math
        __syncthreads();
        const int k0 = kb * OB_FL_BN;
        if (k0 > wave_last_q) continue;         // (wave-uniform) every key of this block is masked for this wave's queries

        // ---- S^T = K . Q^T: 4 key tiles x 2 query tiles, K = D
        ob_float4 sc[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) sc[kt][qt] = (ob_float4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < DK; ++ds) {
                const ob_half8 a = *reinterpret_cast<const ob_half8 *>(&Ks[16 * kt + lr][32 * ds + 8 * g]);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) sc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[qt][ds], sc[kt][qt], 0, 0, 0);
            }
        }
        // ---- causal mask (only blocks that reach this wave's diagonal or the end of the keys), online softmax
        const bool diag = k0 + OB_FL_BN - 1 > A.past + m0 + 32 * wave || k0 + OB_FL_BN > L;
        ob_half8 pb[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (diag) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kidx = k0 + 16 * kt + 4 * g + e;
                        if (kidx > qpos[qt] || kidx >= L) sc[kt][qt][e] = -INFINITY;
                    }
            }
            float mx = sc[0][qt][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sc[kt][qt][e]);
            mx = ob_fl_col_max(mx);
            const float m_new = fmaxf(m_run[qt], mx);
            // a query row with every key masked so far (padding rows beyond S): keep exp2 arguments finite
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_use) * A.scale_log2e);
            const float nm = -m_use * A.scale_log2e;
            float ls = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][qt][e], A.scale_log2e, nm));
                    sc[kt][qt][e] = p;
                    ls += p;
                }
            l_run[qt] = l_run[qt] * alpha + ls;
            // the running output is rescaled only when some query of the tile saw a new maximum (wave-uniform branch):
            // after the first few key blocks most steps skip these 4 * DT multiplications
            if (__builtin_amdgcn_ballot_w64(m_new != m_run[qt]) != 0) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc_o[dt][qt] *= alpha;
            }
            m_run[qt] = m_new;
            // P^T operands: k-elements of step ks = this lane's 4 keys of tile 2 ks, then of tile 2 ks + 1
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pb[qt][ks][e] = (_Float16)sc[2 * ks][qt][e];
                    pb[qt][ks][4 + e] = (_Float16)sc[2 * ks + 1][qt][e];
                }
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const ob_half8 a = *reinterpret_cast<const ob_half8 *>(&Vt[16 * dt + lr][(32 * ks + 8 * g) ^ (((2 * dt + (lr >> 3)) & 7) << 3)]);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) acc_o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[qt][ks], acc_o[dt][qt], 0, 0, 0);
            }
    }

    // ---- normalise and write: lane holds d = 16 dt + 4 g .. + 3 of query lr of each tile
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int s = m0 + 32 * wave + 16 * qt + lr;
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
}
