// NOT part of the product build (moved out of onebit_amd/csrc in round 3: measured slower than the single-role
// kernel, 909 vs 975 tok/s).  Kept for reference; to try it again, include it after
// ob_decode.h in onebit_hip.hip and restore the launcher from git history (commit 3b2a16b).
//
// Role-split decode GEMV (batch 1, integer sign path, aligned shapes): the kernel onebit_decode_step
// launches for q|k|v, gate|up and down.
//
// Why two roles.  With every CU streaming, a CU can keep only ~64 missed cache lines (8 KB: one
// 16-byte load per lane of 8 waves) in flight; a wave that asks for more HBM data is parked in the
// ISSUE of the load until earlier lines return (tools/burst_probe.hip: the second weight load of a
// wave takes ~900 cycles to issue, six take ~3500, and issuing them one after the other is as fast
// as all at once).  A wave that streams weights therefore cannot also run the prologue -- in the
// single-role kernel (ob_decode.h) the LayerNorm / RMSNorm / quantisation chain and the weight stream
// ran one after the other although each had the machine to itself half of the time.  Here a
// 1024-thread workgroup (one per CU) holds
//   waves 0..7   "prologue waves": fetch the (L2-resident) input vectors, LayerNorm / residual /
//                RMSNorm / SiLU*up in the reference's op order, a_p = fp16(x * h_p), fixed-point
//                digits of wave w's K chunks -> LDS, then flag[w];
//   waves 8..15  "matrix waves": request their packed rows at once (parked in the issue most of the
//                time -- that is the HBM stream), wait for flag[w] of the partner wave that owns the
//                same K chunks, multiply (v_mfma_i32_16x16x64_i8, A operand = w & mask), publish
//                scaled fp32 partials;
// one workgroup barrier, then the first MT*16 threads finish rows (fixed-order sum over the 8 matrix
// waves, fp16 roundings of bitnet.py:115-116, per-tile LayerNorm partials for the next launch).
// Synchronisation inside the workgroup is LDS-only: flags / an arrival counter polled with ds_read
// (the prologue waves' RMSNorm reduction must not wait for the parked matrix waves, so it cannot be
// an s_barrier); LDS operations of a wave complete in order, a flag is stored after
// s_waitcnt lgkmcnt(0) on the data stores.  Arithmetic is identical to ob_dec_gemv_kernel<MATH = 1>.
#pragma once
#include "ob_decode.h"

#define OB_DEC2_THREADS 1024

__device__ __forceinline__ unsigned ob_lds_peek(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ob_lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Sum of NV values over the 8 prologue waves without a workgroup barrier: partials to `slot`
// ([NV][8] floats), one arrival count, fixed summation order (deterministic).
template <int NV>
__device__ __forceinline__ void ob_pro_sum_n(float (&v)[NV], float *slot, unsigned *counter, int w, int lane)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ob_wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) slot[i * 8 + w] = v[i];
        ob_lds_drain();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    while (ob_lds_peek(counter) < 8u) __builtin_amdgcn_s_sleep(0);
    ob_lds_drain();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const ob_float4 a = *reinterpret_cast<const volatile ob_float4 *>(slot + i * 8);
        const ob_float4 b = *reinterpret_cast<const volatile ob_float4 *>(slot + i * 8 + 4);
        v[i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
    }
}

// dynamic LDS: digits [NPROJ][Kpad * 4] | partials [MT][8 waves][16 rows][4 digits] f32 |
//              reduction slots [4][8] f32 | per-wave info [8][NPROJ][8] i32 | flags [32] u32
template <int KV, int MS, int PRO, int NPROJ, bool PST>
__global__ __launch_bounds__(OB_DEC2_THREADS) void ob_dec_gemv2_kernel(const ObGemvArgs A)
{
    constexpr int MT = MS * NPROJ;
    const ObProj PP[3] = {A.p[0], A.p[NPROJ > 1 ? 1 : 0], A.p[NPROJ > 2 ? 2 : 0]};
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef OB_PROFILE_ABLATE
    if (A.ablate == 4) return;
#endif
#ifdef OB_PROFILE_STAMPS
    unsigned long long stamp_[16] = {};
#define OB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp_[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define OB_STAMP_FLUSH() do { if (A.dbg && (threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) A.dbg[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + i_] = stamp_[i_]; } } while (0)
#else
#define OB_STAMP(i) do { } while (0)
#define OB_STAMP_FLUSH() do { } while (0)
#endif
    OB_STAMP(0);
    const int K = A.K;
    const int Kpad = (K + 511) & ~511;
    const int nchunks = Kpad >> 9;
    char *lds_q = smem;
    float *lds_red = reinterpret_cast<float *>(smem + (size_t)NPROJ * Kpad * 4);
    float *red = lds_red + MT * 8 * 64;
    int *info = reinterpret_cast<int *>(red + 32);
    unsigned *flags = reinterpret_cast<unsigned *>(info + 8 * 3 * 8);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w = wv & 7;
    const bool is_pro = wv < 8;
    const int gq = lane >> 4;
    const int G = gridDim.x;
    const int per_tile = (nchunks - w + 7) / 8;          // K chunks w, w + 8, ... of a tile belong to wave pair w

    if (tid < 32) flags[tid] = 0u;
    __syncthreads();                                     // every wave is here within cycles of its start

    // slot -> tile
    int trow[MT];
    bool tval[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int ti = (j / NPROJ) * G + (int)blockIdx.x;
        const int ntile = (PP[j % NPROJ].N + 15) >> 4;
        tval[j] = ti < ntile;
        trow[j] = (tval[j] ? ti : 0) << 4;
    }

    if (!is_pro) {
        // ================================ matrix waves =========================================
        // The packed rows are requested only once the partner's input vectors are in (OB_DEC2_GATE: 1 =
        // its loads are issued, 2 = the normalised inputs have arrived, 3 = all of them have): HBM misses
        // queued ahead of the L2-resident vector lines would hold those back by the full HBM latency,
        // and the prologue chain, not the weight stream, is the long pole of the launch.
#ifndef OB_DEC2_GATE
#define OB_DEC2_GATE 2
#endif
        if (OB_DEC2_GATE) {
            while (ob_lds_peek(flags + 16 + w) < (unsigned)OB_DEC2_GATE) __builtin_amdgcn_s_sleep(0);
        }
        ob_u32x4 wreg[MT][KV];
#pragma unroll
        for (int g = 0; g < MS * KV; ++g) {              // MFMA order: group (slot s, chunk ci) = NPROJ loads
#pragma unroll
            for (int p = 0; p < NPROJ; ++p) {
                const int s = g / KV, ci = g % KV, j = s * NPROJ + p;
#ifdef OB_PROFILE_ABLATE
                if (A.ablate == 2 || A.ablate == 3) { wreg[j][ci] = (ob_u32x4){0x12345678u + lane, 0x9abcdef0u, 0x0f1e2d3cu, 0x55aa55aau}; continue; }
#endif
                wreg[j][ci] = ob_dec_load_w<true>(PP[p].w, PP[p].N, K, PP[p].ldw, trow[j], min(w + ci * 8, nchunks - 1), lane);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        OB_STAMP(1);
        while (ob_lds_peek(flags + w) == 0u) __builtin_amdgcn_s_sleep(1);
        ob_lds_drain();
        OB_STAMP(6);
        const int cpc = lane & 3;
        int sdig[NPROJ];
        float fsc[NPROJ];
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) {
            const volatile int *ip = info + (w * 3 + p) * 8;
            sdig[p] = ip[cpc];
            const int e = ip[4];
            const bool nonfinite = ip[5] != 0;
            // 2^(8c) / (128 * 2^(22-e)); a non-finite activation makes the whole output row NaN (reference GEMM)
            fsc[p] = nonfinite ? __builtin_nanf("") : __uint_as_float((uint32_t)(127 + 8 * cpc) << 23) * __uint_as_float((uint32_t)(e - 29 + 127) << 23);
        }
        ob_i32x4 acc[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[j] = (ob_i32x4){0, 0, 0, 0};
        const char *bq = lds_q + (size_t)(gq * 16 + cpc) * 32;
#pragma unroll
        for (int g = 0; g < MS * KV; ++g) {
            const int s = g / KV, ci = g % KV;
            if (ci < per_tile) {
                const int ch = w + ci * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
                        ob_i32x4 bv[NPROJ];
#pragma unroll
                        for (int p = 0; p < NPROJ; ++p)
                            bv[p] = *reinterpret_cast<const ob_i32x4 *>(bq + (size_t)p * Kpad * 4 + (size_t)ch * 2048 + q * 128 + jh * 16);
#pragma unroll
                        for (int p = 0; p < NPROJ; ++p) {
                            const int j = s * NPROJ + p;
                            const uint32_t ww = wreg[j][ci][q];
                            ob_i32x4 av;
#pragma unroll
                            for (int v = 0; v < 4; ++v) av[v] = (int)(ww & (0x01010101u << (4 * jh + v)));
                            acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv[p], acc[j], 0, 0, 0);
                        }
#ifdef OB_PROFILE_STAMPS
                        if (g == 0 && q == 0 && jh == 0) OB_STAMP(7);
#endif
                    }
                }
            }
        }
        OB_STAMP(8);
        // per wave, row and digit c: (S_c - 2 B_c) exact in int32 -> fp32, scaled by exact powers of two
        if ((lane & 15) < 4) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float *dst = lds_red + (((j * 8 + w) * 16 + 4 * gq) << 2) + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[4 * r] = (float)(sdig[j % NPROJ] - 2 * acc[j][r]) * fsc[j % NPROJ];
            }
        }
        __syncthreads();
        OB_STAMP(9);
        OB_STAMP_FLUSH();
        return;
    }

    // ================================== prologue waves ==========================================
    // element ownership: thread tid < 512 holds elements (v * 512 + tid) * 8 .. + 7 = chunk (v * 8 + w)
    bool valid[KV];
    int vbase[KV];
    ob_half8 v0[KV], v1[KV], v2[KV], hp[NPROJ][KV];
    _Float16 c0h = (_Float16)0, c1h = (_Float16)0;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
        const int base = (v * 512 + tid) * 8;
        valid[v] = base < K;
        vbase[v] = valid[v] ? base : 0;
    }
    // loads in the order of need: normalised inputs and their statistics first, scale vectors after
#pragma unroll
    for (int v = 0; v < KV; ++v) {
        if (PRO == OB_P_PLAIN) v0[v] = ob_ld8<false>(A.xin + vbase[v]);
        else if (PRO == OB_P_SWIGLU) { v0[v] = ob_ld8<false>(A.u_gate + vbase[v]); v1[v] = ob_ld8<false>(A.u_up + vbase[v]); }
        else if (PRO == OB_P_RES_LN_RMS) { v0[v] = ob_ld8<false>(A.u_prev + vbase[v]); v1[v] = ob_ld8<false>(A.hres_in + vbase[v]); }
    }
    ObTileStats<KV> ts0, ts1;
    if (PST) {
        if (PRO == OB_P_SWIGLU) { ob_tiles_load<KV>(ts0, A.st_gate, lane); ob_tiles_load<KV>(ts1, A.st_up, lane); }
        if (PRO == OB_P_RES_LN_RMS) ob_tiles_load<KV>(ts0, A.st_prev, lane);
    } else {
        if (PRO == OB_P_SWIGLU) { c0h = A.u_gate[0]; c1h = A.u_up[0]; }
        if (PRO == OB_P_RES_LN_RMS) c0h = A.u_prev[0];
    }
    if (PRO == OB_P_EMBED_RMS) {
        const _Float16 *row = A.embed + (int64_t)(*A.token) * K;
#pragma unroll
        for (int v = 0; v < KV; ++v) v1[v] = ob_ld8<false>(row + vbase[v]);
    }
#pragma unroll
    for (int v = 0; v < KV; ++v) {
        if (PRO == OB_P_EMBED_RMS || PRO == OB_P_RES_LN_RMS) v2[v] = ob_ld8<false>(A.rms_w + vbase[v]);
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) hp[p][v] = ob_ld8<false>(PP[p].h + vbase[v]);
    }
    // epilogue scale of the row this thread finishes (branch-free selection, one unconditional load)
    const int jo = min(tid >> 4, MT - 1);
    int trow_o = trow[0];
    bool tval_o = tval[0];
#pragma unroll
    for (int j = 1; j < MT; ++j) {
        trow_o = jo == j ? trow[j] : trow_o;
        tval_o = jo == j ? tval[j] : tval_o;
    }
    const int p_out = jo - (jo / NPROJ) * NPROJ;
    const _Float16 *g_ptr = PP[0].g;
    _Float16 *u_out = PP[0].u;
    float *st_sel = PP[0].st;
    int N_o = PP[0].N;
#pragma unroll
    for (int p = 1; p < NPROJ; ++p) {
        g_ptr = p_out == p ? PP[p].g : g_ptr;
        u_out = p_out == p ? PP[p].u : u_out;
        st_sel = p_out == p ? PP[p].st : st_sel;
        N_o = p_out == p ? PP[p].N : N_o;
    }
    const int n_raw = trow_o + (tid & 15);
    const bool fin = (tid < MT * 16) && tval_o && n_raw < N_o;
    const int n_out = min(n_raw, N_o - 1);
    const _Float16 g_h = g_ptr[n_out];
    float *st_out = tval_o ? st_sel : nullptr;
    const int tile_out = trow_o >> 4;
    __builtin_amdgcn_sched_barrier(0);
    if (OB_DEC2_GATE == 1 && lane == 0) __hip_atomic_store(flags + 16 + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    OB_STAMP(1);
#define OB_GATE_OPEN(level) do { if (OB_DEC2_GATE == (level) && lane == 0) __hip_atomic_store(flags + 16 + w, (unsigned)(level), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)

    // ---- prologue arithmetic, the reference's op order and rounding points (see ob_decode.h) --------
    ob_half8 xh[KV];
    if (PRO == OB_P_PLAIN) {
#pragma unroll
        for (int v = 0; v < KV; ++v) xh[v] = v0[v];
        asm volatile("" :: "v"(xh[0]));
        OB_GATE_OPEN(2);
    } else if (PRO == OB_P_SWIGLU) {
        float mg, rg, mu, ru;
        if (PST) {
            ob_tiles_combine<KV>(ts0, K, A.ln_eps, lane, mg, rg);
            ob_tiles_combine<KV>(ts1, K, A.ln_eps, lane, mu, ru);
        } else {
            const float c0 = (float)c0h, c1 = (float)c1h;
            ob_float2 sg2 = {0.f, 0.f}, qg2 = {0.f, 0.f}, su2 = {0.f, 0.f}, qu2 = {0.f, 0.f};
#pragma unroll
            for (int v = 0; v < KV; ++v)
                if (valid[v]) { ob_stats8(v0[v], c0, sg2, qg2); ob_stats8(v1[v], c1, su2, qu2); }
            float s[4] = {sg2[0] + sg2[1], qg2[0] + qg2[1], su2[0] + su2[1], qu2[0] + qu2[1]};
            ob_pro_sum_n<4>(s, red, flags + 8, w, lane);
            ob_ln_stats(s[0], s[1], c0, K, A.ln_eps, mg, rg);
            ob_ln_stats(s[2], s[3], c1, K, A.ln_eps, mu, ru);
        }
        OB_STAMP(2);
        OB_GATE_OPEN(2);
        const float ng = -mg * rg, nu = -mu * ru;
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            ob_half8 sg, up;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const _Float16 gh = ob_ln_apply_h(v0[v][i], rg, ng);
                up[i] = ob_ln_apply_h(v1[v][i], ru, nu);
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf((float)gh, -1.44269504088896341f, 0.0f));
                sg[i] = (_Float16)__builtin_fmaf((float)gh, __builtin_amdgcn_rcpf(1.0f + e), 0.0f);
            }
            xh[v] = sg * up;                                    // act_fn(gate) * up, modeling_bitllama.py:257
        }
    } else {
        ob_half8 hv[KV];
        if (PRO == OB_P_RES_LN_RMS) {
            float mean, rstd;
            if (PST) {
                ob_tiles_combine<KV>(ts0, K, A.ln_eps, lane, mean, rstd);
            } else {
                const float c0 = (float)c0h;
                ob_float2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                for (int v = 0; v < KV; ++v)
                    if (valid[v]) ob_stats8(v0[v], c0, s2, q2);
                float s[2] = {s2[0] + s2[1], q2[0] + q2[1]};
                ob_pro_sum_n<2>(s, red, flags + 8, w, lane);
                ob_ln_stats(s[0], s[1], c0, K, A.ln_eps, mean, rstd);
            }
            OB_STAMP(2);
            OB_GATE_OPEN(2);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                ob_half8 ln;
#pragma unroll
                for (int i = 0; i < 8; ++i) ln[i] = ob_ln_apply_h(v0[v][i], rstd, nmr);
                hv[v] = v1[v] + ln;                 // residual + hidden_states, modeling_bitllama.py:912,918
            }
        } else {
#pragma unroll
            for (int v = 0; v < KV; ++v) hv[v] = v1[v];
            asm volatile("" :: "v"(hv[0]));
            OB_GATE_OPEN(2);
        }
        float ss[1] = {0.f};
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            if (valid[v]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const ob_half2 pr = {hv[v][2 * i], hv[v][2 * i + 1]};
                    ss[0] = __builtin_amdgcn_fdot2(pr, pr, ss[0], false);
                }
            }
        }
        ob_pro_sum_n<1>(ss, red + 16, flags + 9, w, lane);
        OB_STAMP(3);
        const float rs = __builtin_amdgcn_rsqf(ss[0] * __builtin_amdgcn_rcpf((float)K) + A.rms_eps);
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            ob_half8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (_Float16)__builtin_fmaf((float)hv[v][i], rs, 0.0f);   // fp16(h * rsqrt)
            xh[v] = v2[v] * t;
            if (blockIdx.x == 0 && A.hres_out && valid[v]) ob_st8<false>(A.hres_out + vbase[v], hv[v]);
        }
    }
    OB_STAMP(4);

    // ---- a_p = fp16(x * h_p), wave-local fixed point (exponent from this wave's largest element) --------
    ob_half8 ah[NPROJ][KV];
    int e_w[NPROJ];
    bool nonfinite[NPROJ];
#pragma unroll
    for (int v = 0; v < KV; ++v)
        if (!valid[v]) xh[v] = (ob_half8)(_Float16)0;
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) {
        ob_u16x2 mx = {0, 0};
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            ah[p][v] = ob_quad_transpose(xh[v] * hp[p][v], lane);
            const ob_u32x4 bits = __builtin_bit_cast(ob_u32x4, ah[p][v]);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                mx = __builtin_elementwise_max(mx, __builtin_bit_cast(ob_u16x2, bits[d] & 0x7fff7fffu));
        }
        const uint32_t m = ob_wave_max_u32(max((uint32_t)mx[0], (uint32_t)mx[1]));
        e_w[p] = (int)max(m >> 10, 1u) - 15;
        nonfinite[p] = m >= 0x7c00u;
    }
    OB_GATE_OPEN(3);                                   // every prologue vector has been consumed
    OB_STAMP(5);
    const int jp = lane & 3;
    const float cj0 = __uint_as_float((uint32_t)(127 + 7 - 2 * jp) << 23);
    const float cj1 = jp < 3 ? __uint_as_float((uint32_t)(127 + 6 - 2 * jp) << 23) : -1.0f;
    const int vj0 = (int)(0x01010101u << (2 * jp)), vj1 = (int)(0x01010101u << (2 * jp + 1));
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) {
        const int e = e_w[p];
        const float scale = __uint_as_float((uint32_t)(22 - e + 127) << 23);          // 2^(22-e)
        const float sc0 = scale * cj0, sc1 = scale * cj1;
        int D[4] = {0, 0, 0, 0};
        char *dst = lds_q + (size_t)p * Kpad * 4 + (lane >> 2) * 128 + jp * 8;
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            uint32_t T[2][4];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint32_t W[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float f = __builtin_fmaf((float)ah[p][v][2 * i + s2], s2 ? sc1 : sc0, 0.0f);
                    int m;
                    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(m) : "v"(f));
                    W[i] = ((uint32_t)m + 0x00808080u) ^ 0x00808080u;
                }
                const uint32_t u0 = __builtin_amdgcn_perm(W[1], W[0], 0x05010400u), u1 = __builtin_amdgcn_perm(W[1], W[0], 0x07030602u);
                const uint32_t w0 = __builtin_amdgcn_perm(W[3], W[2], 0x05010400u), w1 = __builtin_amdgcn_perm(W[3], W[2], 0x07030602u);
                T[s2][0] = __builtin_amdgcn_perm(w0, u0, 0x05040100u);
                T[s2][1] = __builtin_amdgcn_perm(w0, u0, 0x07060302u);
                T[s2][2] = __builtin_amdgcn_perm(w1, u1, 0x05040100u);
                T[s2][3] = __builtin_amdgcn_perm(w1, u1, 0x07060302u);
#pragma unroll
                for (int c = 0; c < 4; ++c) D[c] = __builtin_amdgcn_sdot4((int)T[s2][c], s2 ? vj1 : vj0, D[c], false);
            }
            if ((v * 8 + w) * 512 < Kpad) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<ob_u32x2 *>(dst + (size_t)(v * 8 + w) * 2048 + c * 32) = (ob_u32x2){T[0][c], T[1][c]};
            }
        }
        // S of this wave, exact, digit (lane & 3) in every lane
        const bool hi2 = lane & 2, hi1 = lane & 1;
        int k0 = hi2 ? D[2] : D[0], k1 = hi2 ? D[3] : D[1];
        const int s0 = hi2 ? D[0] : D[2], s1 = hi2 ? D[1] : D[3];
        k0 += __builtin_amdgcn_update_dpp(0, s0, 0x4E, 0xF, 0xF, false);
        k1 += __builtin_amdgcn_update_dpp(0, s1, 0x4E, 0xF, 0xF, false);
        int t = hi1 ? k1 : k0;
        const int sd = hi1 ? k0 : k1;
        t += __builtin_amdgcn_update_dpp(0, sd, 0xB1, 0xF, 0xF, false);
        t += __builtin_amdgcn_update_dpp(0, t, 0x124, 0xF, 0xF, false);
        t += __builtin_amdgcn_update_dpp(0, t, 0x128, 0xF, 0xF, false);
        auto s16 = __builtin_amdgcn_permlane16_swap((uint32_t)t, (uint32_t)t, false, false);
        t = (int)(s16[0] + s16[1]);
        auto s32 = __builtin_amdgcn_permlane32_swap((uint32_t)t, (uint32_t)t, false, false);
        const int sdig = (int)(s32[0] + s32[1]);
        volatile int *ip = info + (w * 3 + p) * 8;
        if (lane < 4) ip[lane] = sdig;
        if (lane == 4) ip[4] = e;
        if (lane == 5) ip[5] = nonfinite[p] ? 1 : 0;
    }
    ob_lds_drain();
    if (lane == 0) __hip_atomic_store(flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    OB_STAMP(6);

    __syncthreads();                                   // all partials of the matrix waves are in LDS
    OB_STAMP(9);
    float uval = 0.f;
    if (fin) {
        const int r = tid & 15;
        float z = 0.f;
#pragma unroll
        for (int m8 = 0; m8 < 8; ++m8) {
            const ob_float4 t = *reinterpret_cast<const ob_float4 *>(lds_red + (((jo * 8 + m8) * 16 + r) << 2));
            z += (t[0] + t[1]) + (t[2] + t[3]);
        }
        const _Float16 uh = (_Float16)(ob_round_h(z) * (float)g_h);   // fp16(z) (bitnet.py:115), * g -> fp16 (:116)
        u_out[n_out] = uh;
        uval = (float)uh;
    }
    if (tid < MT * 16) {
        const float sm = ob_row16_sum(fin ? uval : 0.f);
        const float dv = fin ? uval - sm * 0.0625f : 0.f;
        const float m2 = ob_row16_sum(dv * dv);
        if (st_out && (tid & 15) == 0) *reinterpret_cast<ob_float2 *>(st_out + 2 * tile_out) = (ob_float2){sm, m2};
    }
    OB_STAMP(10);
    OB_STAMP_FLUSH();
#undef OB_STAMP
#undef OB_STAMP_FLUSH
#undef OB_GATE_OPEN
}
