// Whole-token decode kernels (batch 1) for gfx950: five launches per decoder layer.
//
//   dec_gemv<qkv>     h = residual stream (+ LayerNorm of the previous down_proj output), RMSNorm,
//                     three 1-bit GEMVs (q, k, v) -> pre-LayerNorm u_q, u_k, u_v
//   dec_attn          LayerNorm(u_q,u_k,u_v) -> RoPE -> KV append -> softmax(QK^T/sqrt d) V per head
//   dec_gemv<o>       1-bit GEMV o_proj -> u_o
//   dec_gemv<gateup>  h += LayerNorm(u_o); RMSNorm; gate and up GEMVs -> u_gate, u_up
//   dec_gemv<down>    silu(LayerNorm(u_gate)) * LayerNorm(u_up); down GEMV -> u_down
//   dec_lmhead        h += LayerNorm(u_down); final RMSNorm; fp16 lm_head GEMV; per-workgroup argmax
//   dec_argmax        greedy token, position += 1 (all state stays on the device: graph replay)
//
// Every LayerNorm of the reference's BitLinearInf (bitnet.py:118) needs statistics over a whole
// output row, i.e. over every workgroup of the producing GEMV; a kernel boundary is the cheapest
// all-to-all synchronisation on this chip (MI355X_MICROARCH.md, price list "boundary"), so the
// producer writes pre-LayerNorm u and every consumer workgroup recomputes the (tiny) statistics
// itself from the L2-resident vector.  Rounding points follow the reference's fp16 tensor ops.
//
// GEMV work decomposition: a persistent grid (one 512-thread workgroup per CU).  The unit of work
// is a 16-row tile of one projection; a workgroup owns tiles b, b+G, b+2G, ...  Its 8 waves split K
// in 512-weight chunks (one global_load_dwordx4 per lane), each chunk = 16 MFMA 16x16x32 with the
// packed words as the A operand.  All weight loads of a wave are issued before the prologue so the
// HBM latency overlaps the statistics; partial sums meet in LDS.
#pragma once
#include <type_traits>
#include "ob_common.h"
#include "ob_rowstats.h"

#ifndef OB_DEC_THREADS
#define OB_DEC_THREADS 512
#endif
#define OB_DEC_WAVES (OB_DEC_THREADS / 64)
#define OB_DEC_MAXV 4            // per-thread vectors of 8 halves: vector widths up to 16384
#ifndef OB_SMFMA
#define OB_SMFMA 1               // integer path: the digit sums S come from the matrix pipe (see the kernel); 0 = v_dot4 + DPP
#endif

// LDS layout of the integer-path decode GEMV, ONE statement for the kernel and for the host's launch size (they were two
// hand-kept formulas): digit image [nproj][KV * waves * 512 elements][4 digit bytes] | cross-wave partials + wave scratch
// (ob_dec_red_off floats) | 256 floats of reduction slots.
__host__ __device__ constexpr int ob_dec_kq(int kv) { return kv * OB_DEC_WAVES * 512; }                      // elements of a wave row set
__host__ __device__ constexpr int ob_dec_red_off(int mt) { return mt * OB_DEC_WAVES * 64 + 3 * OB_DEC_WAVES * 16 + 16; }
__host__ __device__ constexpr size_t ob_dec_lds_i8_bytes(int nproj, int kv, int mt)
{
    return (size_t)nproj * ob_dec_kq(kv) * 4 + ((size_t)ob_dec_red_off(mt) + 256) * 4;
}

struct ObProj {
    const uint32_t *w;           // packed signs [N, ldw words]
    const _Float16 *h;           // input_factor [K]
    const _Float16 *g;           // weight_scale [N]
    _Float16 *u;                 // out: pre-LayerNorm u [N]
    float *st;                   // out (optional): per 16-row tile (sum, M2) of u, interleaved [ceil(N/16)][2]
    int N, K, ldw;
};

enum ObPrologue { OB_P_PLAIN = 0, OB_P_EMBED_RMS = 1, OB_P_RES_LN_RMS = 2, OB_P_SWIGLU = 3 };

struct ObGemvArgs {
    ObProj p[3];
    int nproj;
    int prologue;
    int K;                         // shared in_features of the projections
    // prologue inputs
    const _Float16 *xin;           // PLAIN: input vector [K]
    const _Float16 *embed;         // EMBED_RMS: embedding table [vocab, K]
    const int *token;              // EMBED_RMS: device token id
    const _Float16 *hres_in;       // RES_LN_RMS: residual stream in [K]
    const _Float16 *u_prev;        // RES_LN_RMS: pre-LN output of the previous projection [K]
    _Float16 *hres_out;            // EMBED_RMS / RES_LN_RMS: residual stream out [K] (workgroup 0 writes)
    const _Float16 *rms_w;         // RMSNorm weight [K]
    const _Float16 *u_gate, *u_up; // SWIGLU: pre-LN gate / up [K]
    // per-tile (sum, M2) pairs written by the producers of u_prev / u_gate / u_up (PST kernels): the
    // LayerNorm statistics are then a wave-local combine of K/16 pairs -- no pass over the vector, no
    // workgroup barrier.  Buffers hold a multiple of 256 tiles (reads beyond K/16 are masked).
    const float *st_prev, *st_gate, *st_up;
    float rms_eps, ln_eps;
    // WGP kernels (one projection per workgroup): workgroups [wg_end[p-1], wg_end[p]) own the tiles of projection p
    int wg_end[3];
    // EMBED_RMS (first launch of a step), optional: workgroup 0 copies the rotary rows of the current position,
    // cos[pos] | sin[pos], to rope_out [2 * rope_D] for the step's attention launches (ObAttnArgs.rope_cur)
    const int *rope_pos;
    const _Float16 *rope_cos, *rope_sin;
    _Float16 *rope_out;
    int rope_D, rope_max;
    int ablate;                    // profiling builds only (-DOB_PROFILE_ABLATE + OB_ABLATE env); 0 = normal
    unsigned long long *dbg;       // profiling builds only: per-workgroup phase timestamps [grid][8]
    // ---- round 5, appended so that every field above keeps its kernarg offset (8 bytes more in the MIDDLE of this struct cost
    //      the layer-order chain 0.37 us per layer: profiles/r05_bench.json vs the same day's baseline) ----
    // RES_LN_RMS, BIAS kernels: bias [K] of the projection that produced u_prev (o_proj with config.attention_bias,
    // modeling_bitllama.py:454): r = hres_in + fp16(LayerNorm(u_prev) + bias_prev)  (bitnet.py:119-120 then :912)
    const _Float16 *bias_prev;
    int zout;                      // host-side selector of the ZOUT instances (fp32 partial sums of a K slice to ObProj.u); PLAIN only
};

// Block-wide sums of NV values through LDS.  `red` must be a slot (16*NV floats, 16-byte aligned)
// not used by any other reduction still in flight, so a single barrier suffices.  NW (waves per
// workgroup) is a compile-time constant: the partials of one value are two ds_read_b128, not a
// loop of dependent scalar LDS reads (that loop cost ~1.7 us per reduction).
template <int NV, int NW>
__device__ __forceinline__ void ob_block_sum_n(float (&v)[NV], float *red)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ob_wave_sum(v[i]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[i * 16 + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < NW; w4 += 4) {
            const ob_float4 a = *reinterpret_cast<const ob_float4 *>(red + i * 16 + w4);
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
        v[i] = s;
    }
}

// Shifted sums of 8 halves, two lanes of packed fp32 (v_pk_add_f32 / v_pk_fma_f32): s += (u - c),
// q += (u - c)^2; even and odd elements accumulate separately and are added at the end.
__device__ __forceinline__ void ob_stats8(const ob_half8 u, float c, ob_float2 &s, ob_float2 &q)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ob_float2 d = {(float)u[2 * i], (float)u[2 * i + 1]};
        d = d - c;
        s += d;
        q = __builtin_elementwise_fma(d, d, q);
    }
}

// LayerNorm statistics (mean, rstd) from shifted sums s1 = sum(u - c), s2 = sum((u - c)^2).
__device__ __forceinline__ void ob_ln_stats(float s1, float s2, float c, int n, float eps, float &mean,
                                            float &rstd)
{
    // hardware reciprocal / reciprocal square root (1 ulp), as GPU LayerNorm kernels use: the IEEE
    // division + sqrt sequences are ~40 dependent instructions on every thread's critical path
    const float inv_n = __builtin_amdgcn_rcpf((float)n);
    const float m1 = s1 * inv_n;
    mean = c + m1;
    const float var = fmaxf(s2 * inv_n - m1 * m1, 0.f);
    rstd = __builtin_amdgcn_rsqf(var + eps);
}

// (ob_ln_apply: ob_rowstats.h)

// The same LayerNorm element as ONE instruction (v_fma_mixlo_f16: fp16 in, fp32 fma, fp16 out):
// fp16(u * rstd + (-mean * rstd)).  One rounding of the normalised value instead of two; the forms
// agree except at fp16 rounding ties of the intermediate (~2^-13 of elements, one ulp), the same
// spread as between LayerNorm implementations of different torch backends.
__device__ __forceinline__ _Float16 ob_ln_apply_h(_Float16 u, float rstd, float nmr)
{
    return (_Float16)__builtin_fmaf((float)u, rstd, nmr);
}

__device__ __forceinline__ float ob_silu_h(float x)   // fp16 silu: fp32 math, one rounding
{
    return ob_round_h(x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)));
}


// ---------------------------------------------------------------------------------------------
// Producer-side LayerNorm partials.  The thread rows that finalise a 16-row tile hold its u values
// in one 16-lane DPP row: they publish (sum, M2 = sum of squared deviations from the tile mean).
// A consumer combines the n/16 pairs with the parallel-variance formula (Chan et al.):
//   mean = sum_i s_i / n,   M2 = sum_i [ M2_i + c_i (s_i / c_i - mean)^2 ],   var = M2 / n (biased)
// -- per wave, redundantly, 4 tiles per lane and 256-tile block: two DPP wave reductions instead of a
// pass over the vector, a shifted one-pass sum and a workgroup barrier.
// ---------------------------------------------------------------------------------------------
template <int NV>
struct ObTileStats { ob_float4 a[NV][2]; };

template <int NV>
__device__ __forceinline__ void ob_tiles_load(ObTileStats<NV> &r, const float *st, int lane)
{
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const ob_float4 *p = reinterpret_cast<const ob_float4 *>(st + (size_t)(v * 64 + lane) * 8);
        r.a[v][0] = p[0];
        r.a[v][1] = p[1];
    }
}

template <int NV>
__device__ __forceinline__ void ob_tiles_combine(const ObTileStats<NV> &r, int n, float eps, int lane, float &mean, float &rstd)
{
    // n % 16 == 0 (host-checked): every tile below n / 16 is full, the rest of a block is masked
    const int ntiles = n >> 4;
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s += (v * 64 + lane) * 4 + i < ntiles ? r.a[v][i >> 1][2 * (i & 1)] : 0.f;
    }
    s = ob_wave_sum(s);
    const float inv_n = __builtin_amdgcn_rcpf((float)n);
    mean = s * inv_n;
    float m2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = __builtin_fmaf(r.a[v][i >> 1][2 * (i & 1)], 0.0625f, -mean);
            const float t = __builtin_fmaf(16.0f * d, d, r.a[v][i >> 1][2 * (i & 1) + 1]);
            m2 += (v * 64 + lane) * 4 + i < ntiles ? t : 0.f;
        }
    }
    m2 = ob_wave_sum(m2);
    rstd = __builtin_amdgcn_rsqf(fmaxf(m2 * inv_n, 0.f) + eps);
}

// Cooperative form for the long vectors of the SwiGLU prologue (two vectors of up to 16384 elements: 11 KB of pairs at
// 7B -- read by each of the 8 waves that was 88 KB per workgroup through the vector L1, more than the vectors
// themselves).  Each WAVE loads and combines 1/8 of the pairs (lanes 0 .. 8 NV - 1, one 4-tile slot each) and publishes
// (elements, mean, M2) of its share; after ONE workgroup barrier every thread merges the 8 triples (Chan et al. again).
struct ObTileSlot { ob_float4 a0, a1; };
template <int NV>
__device__ __forceinline__ void ob_tiles_slot_load(ObTileSlot &r, const float *st, int wave, int lane)
{
    const int q = wave * (8 * NV) + min(lane, 8 * NV - 1);
    const ob_float4 *p = reinterpret_cast<const ob_float4 *>(st + (size_t)q * 8);
    r.a0 = p[0];
    r.a1 = p[1];
}
template <int NV>
__device__ __forceinline__ ob_float4 ob_tiles_slot_partial(const ObTileSlot &r, int n, int wave, int lane)
{
    const int ntiles = n >> 4;
    const int t0 = (wave * (8 * NV) + lane) * 4;
    const bool act = lane < 8 * NV;
    const float sv[4] = {r.a0[0], r.a0[2], r.a1[0], r.a1[2]}, qv[4] = {r.a0[1], r.a0[3], r.a1[1], r.a1[3]};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (act && t0 + i < ntiles) ? sv[i] : 0.f;
    s = ob_wave_sum(s);
    const int tiles_w = min(max(ntiles - wave * (32 * NV), 0), 32 * NV);      // this wave's tiles that exist
    const float n_w = 16.0f * (float)tiles_w;
    const float mean_w = s * __builtin_amdgcn_rcpf(fmaxf(n_w, 1.0f));
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float d = __builtin_fmaf(sv[i], 0.0625f, -mean_w);
        m2 += (act && t0 + i < ntiles) ? __builtin_fmaf(16.0f * d, d, qv[i]) : 0.f;
    }
    m2 = ob_wave_sum(m2);
    return (ob_float4){n_w, mean_w, m2, 0.f};
}
// merge of the 8 per-wave triples at slot[w * 8 + 4 * vec .. + 3] (floats)
template <int NW>
__device__ __forceinline__ void ob_tiles_slot_merge(const float *slot, int vec, int n, float eps, float &mean, float &rstd)
{
    ob_float4 t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = *reinterpret_cast<const ob_float4 *>(slot + w * 8 + 4 * vec);
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) S = __builtin_fmaf(t[w][0], t[w][1], S);
    const float inv_n = __builtin_amdgcn_rcpf((float)n);
    mean = S * inv_n;
    float m2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float d = t[w][1] - mean;
        m2 += __builtin_fmaf(t[w][0] * d, d, t[w][2]);
    }
    rstd = __builtin_amdgcn_rsqf(fmaxf(m2 * inv_n, 0.f) + eps);
}

// (ObTileStatsRt / ob_tiles_load_rt / ob_tiles_combine_rt: ob_rowstats.h)

// sum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ float ob_row16_sum(float v)
{
    v += OB_DPP_F(v, 0xB1, 0xF);
    v += OB_DPP_F(v, 0x4E, 0xF);
    v += OB_DPP_F(v, 0x141, 0xF);
    v += OB_DPP_F(v, 0x140, 0xF);
    return v;
}

// 4x4 dword transpose inside each quad of lanes: lane l, dword d  <-  lane d, dword l.  With the
// contiguous element ownership (lane 4Q + i holds elements 32Q + 8i .. + 7 as 4 fp16 pairs) this
// yields the strided ownership of the integer path (lane (Q, jp): elements 32Q + 8i + 2jp + s), so
// the prologue vectors are fetched with ONE 16-byte load per lane instead of four 4-byte ones
// (the texture addresser, not the data, bounded the load phase: 4x the instructions for the same
// cache lines).  Two butterfly stages (lane ^ 2 then lane ^ 1), 16 VALU ops.
__device__ __forceinline__ ob_half8 ob_quad_transpose(const ob_half8 x, int lane)
{
    ob_u32x4 d = __builtin_bit_cast(ob_u32x4, x);
    const bool hi = lane & 2, odd = lane & 1;
    {
        const uint32_t s0 = hi ? d[0] : d[2], s1 = hi ? d[1] : d[3];
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0x4E, 0xF, 0xF, false);
        const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0x4E, 0xF, 0xF, false);
        d[0] = hi ? r0 : d[0]; d[1] = hi ? r1 : d[1];
        d[2] = hi ? d[2] : r0; d[3] = hi ? d[3] : r1;
    }
    {
        const uint32_t s0 = odd ? d[0] : d[1], s2 = odd ? d[2] : d[3];
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);
        const uint32_t r2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s2, 0xB1, 0xF, 0xF, false);
        d[0] = odd ? r0 : d[0]; d[1] = odd ? d[1] : r0;
        d[2] = odd ? r2 : d[2]; d[3] = odd ? d[3] : r2;
    }
    return __builtin_bit_cast(ob_half8, d);
}

// 16 MFMAs for one 512-weight chunk of a 16-row tile.  w4: this lane's 4 packed words (row = lane&15,
// k = 128*(lane>>4) + 32*q + bit).  a: LDS activations of the chunk.  Every column of the B
// operand carries the same token (T = 1), so no masking is needed: all 16 result columns agree.
__device__ __forceinline__ void ob_dec_chunk(const ob_u32x4 w4, const _Float16 *a_chunk, int gq, ob_float4 &acc)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t w = w4[q];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            uint32_t e[8];
            ob_expand16((w >> (16 * hf)) & 0xffffu, e);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const ob_half8 b = *reinterpret_cast<const ob_half8 *>(a_chunk + gq * 128 + q * 32 + (2 * hf + s2) * 8);
                ob_u32x4 av = {e[4 * s2 + 0], e[4 * s2 + 1], e[4 * s2 + 2], e[4 * s2 + 3]};
                ob_half8 aop;
                __builtin_memcpy(&aop, &av, 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, b, acc, 0, 0, 0);
            }
        }
    }
}

// One lane's 4 packed words of (row0 + lane&15, chunk); branch-free (indices are clamped into the
// row and out-of-range words zeroed afterwards) so that the load can be issued ahead of everything.
// ALIGNED: K % 128 == 0 and 16-byte aligned rows: the 4 words are all inside or all outside the row
// and one dwordx4 does it (non-temporal: each word is read exactly once per token).
template <bool ALIGNED>
__device__ __forceinline__ ob_u32x4 ob_dec_load_w(const uint32_t *w, int N, int K, int ldw, int row0, int chunk, int lane)
{
    const int r = lane & 15, gq = lane >> 4;
    const int row = min(row0 + r, N - 1);
    const int word = chunk * 16 + gq * 4;
    const int nwords = K >> 5;
    const uint32_t *rowp = w + (int64_t)row * ldw;
    ob_u32x4 w4;
    if (ALIGNED) {
        const int wc = min(word, nwords - 4);
#ifndef OB_DEC_W_NT
#define OB_DEC_W_NT 1                           // 0: allocating loads (A/B build for the L2-residency experiment)
#endif
        if (OB_DEC_W_NT) w4 = __builtin_nontemporal_load(reinterpret_cast<const ob_u32x4 *>(rowp + wc));
        else w4 = *reinterpret_cast<const ob_u32x4 *>(rowp + wc);
        // words beyond the row (word >= nwords: the tail of the last chunk, chunks beyond K) are NOT zeroed: they re-read
        // valid words (clamped address) and multiply activations that are zero there (the LDS images are zero-padded), so
        // their value never matters -- and a select on the loaded registers right here is a USE at the issue point: in a
        // straight-line phase it put s_waitcnt vmcnt(0) behind every rolling load (round 4)
        (void)nwords;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t v = rowp[min(word + q, nwords - 1)];
            w4[q] = (word + q < nwords) ? v : 0u;
        }
    }
    return w4;
}

// ---------------------------------------------------------------------------------------------
// Integer sign path (MATH == 1).  Each wave quantises the elements of a = fp16(x*h) it owns (its
// 512-weight chunks) to fixed point relative to their largest one, q = a * 2^(22 - e) (|q| < 2^23;
// exact for every element within 12 binades of that maximum, otherwise rounded at 2^-23 of it --
// below fp32 accumulation noise), and splits q into 4 signed int8 digits.
// The A operand of v_mfma_i32_16x16x64_i8 is then simply  w & (0x01010101 << j)  : byte i of that
// dword is bit (8i + j) of the packed word times 2^j, i.e. ONE v_and per 4 weights.  The factor
// 2^j (and the sign of 0x80 for j = 7) is folded into the activation side: element k with bit
// position j = k % 8 is stored as m' = q * 2^(7-j) (j < 7) or -q (j = 7), so every product is
// 128 * b * q.  With B = sum over set bits (int32 per digit) and S = sum over all elements
// (v_dot4 per lane + an exact cross-lane reduction), S_c - 2 B_c is exact per digit c; one
// conversion to fp32 per (wave, row, digit), scaled by exact powers of two, a fixed-order sum.
// The 4 digits of an element sit in one dword [d0 d1 d2 d3]; the MFMA B operand wants the same
// digit of 4 elements per dword -- with the strided element ownership below that is a
// thread-local 4x4 byte transpose (8 v_perm per 4 elements).
//
// LDS image per projection: [Q = k/32][digit c][32 bytes], the 32 bytes = dwords j = 0..7, dword j
// = digit c of k = 32Q + 8i + j for i = 0..3.  MFMA step (q, jh) of chunk ch, lane (g, c): 16 bytes
// at ((ch*16 + g*4 + q) * 4 + (c & 3)) * 32 + jh * 16.  Lanes with c >= 4 replicate digit c & 3
// (their result columns are simply not used).  A wave reads back only what it wrote itself.
// ---------------------------------------------------------------------------------------------

// Element ownership in the GEMV prologue.  Contiguous (fp16 path): the thread holds 8 consecutive
// elements.  Strided (integer path): lane (Q = lane >> 2, jp = lane & 3) of the wave that owns a
// 512-element chunk holds elements 32Q + 8i + 2jp + s (i = 0..3, s = 0..1; vector index 2i + s), i.e.
// for each of its two bit positions j = 2jp + s the four elements whose sign bits are the four BYTES
// of one masked weight word -- so the digit-planar MFMA operand is a thread-local 4x4 byte transpose.
template <bool STRIDED>
__device__ __forceinline__ ob_half8 ob_ld8(const _Float16 *p)
{
    if (!STRIDED) return *reinterpret_cast<const ob_half8 *>(p);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    const ob_u32x4 r = {q[0], q[4], q[8], q[12]};
    return __builtin_bit_cast(ob_half8, r);
}
template <bool STRIDED>
__device__ __forceinline__ void ob_st8(_Float16 *p, const ob_half8 v)
{
    if (!STRIDED) { *reinterpret_cast<ob_half8 *>(p) = v; return; }
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
    const ob_u32x4 r = __builtin_bit_cast(ob_u32x4, v);
    q[0] = r[0]; q[4] = r[1]; q[8] = r[2]; q[12] = r[3];
}

// ---------------------------------------------------------------------------------------------
// The fused decode GEMV kernel.
//   KV      = ceil(K / 4096): 8-half vectors per thread in the prologue AND 512-weight chunks per
//             wave per tile
//   MS      = tile slots per projection per workgroup; slot j = (s, p) with p = j % NPROJ the
//             projection and s = j / NPROJ: the workgroup owns tile s*G + blockIdx.x of projection p.
//             Projection of a slot is therefore a compile-time constant, and all slots of one
//             projection share their activation (B operand) reads.
//   ALIGNED = K % 128 == 0 and 16-byte aligned packed rows
//   PRO     = prologue, NPROJ = number of projections, MATH: 0 = fp16 MFMA with in-register sign
//             expansion, 1 = integer path (needs ALIGNED)
// All compile-time, so the in-flight registers are statically indexed and the load phase is
// straight-line code: every global load of the kernel (prologue vectors, epilogue scales, weights)
// is issued before the first use of any of them -- one memory round trip, not one per stage --
// with the prologue vectors first (loads return in order) so that the prologue runs underneath the
// weight stream.
// dynamic LDS: activations (fp16: 2 B/k, i8: 4 digit bytes/k) per projection | cross-wave partials |
//              reduction slots (256 floats)
// ---------------------------------------------------------------------------------------------
// WGP (round 4; NPROJ == 1 in the template, A.nproj projections in the launch): ONE PROJECTION PER WORKGROUP.  The
// workgroups of a q|k|v or gate|up launch are dealt to the projections (A.wg_end) and a workgroup's MS slots are all
// tiles of ITS projection.  The per-slot form above gave every workgroup one tile of every projection, so every
// workgroup quantised x * h_p for every p (q|k|v: three amax + digit passes, 359 of 1129 instructions per wave) and
// read every projection's digit planes from LDS for ONE use each (a B operand of v_mfma_i32_16x16x64_i8 is 1 KB per
// wave and step; at 128 B/clk the q|k|v launch spent ~1500 cycles per workgroup on those reads: the gap between
// "digits in LDS" and "first MFMA" in tools/phase_probe.py).  With one projection per workgroup the prologue quantises
// once and each B operand feeds MS MFMAs.
// ZOUT (round 5, K-sharded decode: config 4): the launch multiplies a K SLICE of its projections (weight pointer, input_factor
// and xin already offset to the slice, K = its width, ldw the full row pitch) and stores the fp32 partial sum z of every row to
// ObProj.u (as float *) -- no rounding, no weight_scale, no LayerNorm partials: the caller all-reduces z across the K shards
// and the consumer applies fp16(fp16(z) * g) (bitnet.py:115-116) to the complete sum.
template <int KV, int MS, bool ALIGNED, int PRO, int MATH, int NPROJ, bool PST, bool WGP = false, bool BIAS = false, bool ZOUT = false>
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_dec_gemv_kernel(const ObGemvArgs A)
{
    static_assert(!WGP || NPROJ == 1, "WGP kernels are instantiated with NPROJ = 1");
    static_assert(!BIAS || PRO == OB_P_RES_LN_RMS, "only the residual prologue adds the producer's bias");
    constexpr int MT = MS * NPROJ;
    // digit sums from the matrix pipe where several projections share the launch (gate|up, q|k|v: measured 7.16 -> 6.66 us
    // and 6.57 -> 6.6 us per launch); a single-projection launch with KV = 3 chunks per wave (down) would double its MFMA
    // count for the same saving and measured slower (7.22 -> 7.50 us): it keeps v_dot4 + DPP
    constexpr bool SMF = OB_SMFMA && (NPROJ >= 2 || (WGP && MS >= 3));
    // PST: LayerNorm statistics of the prologue inputs come from the producers' per-tile partials
#ifdef OB_STRIDED_LOADS                     // A/B switch (tools/phase_probe.py): 4-byte strided prologue loads, no transpose
    constexpr bool SD = MATH == 1;
#else
    constexpr bool SD = false;              // prologue vectors are always fetched contiguously (one 16-byte
                                            // load per lane); the integer path transposes inside quads later
#endif
    // the projection descriptors live in SGPRs; selection by slot is compile-time
    int wg_local = (int)blockIdx.x, wg_count = (int)gridDim.x;
    int pw = 0;
    if (WGP) {
        const int b = (int)blockIdx.x;
        pw = (b >= A.wg_end[0] ? 1 : 0) + (b >= A.wg_end[1] ? 1 : 0);
        const int b0 = pw == 0 ? 0 : (pw == 1 ? A.wg_end[0] : A.wg_end[1]);
        wg_count = (pw == 0 ? A.wg_end[0] : (pw == 1 ? A.wg_end[1] : A.wg_end[2])) - b0;
        wg_local = b - b0;
    }
    // every prologue argument is requested in the first scalar-load clause (hipcc fetches kernel arguments where they are
    // first used: a second s_load / s_waitcnt round trip stood in front of the vector requests)
    if (PRO == OB_P_RES_LN_RMS) asm volatile("" :: "s"(A.hres_in), "s"(A.u_prev), "s"(A.hres_out), "s"(A.rms_w), "s"(A.st_prev), "s"(A.K));
    if (BIAS) asm volatile("" :: "s"(A.bias_prev));
    if (PRO == OB_P_EMBED_RMS) asm volatile("" :: "s"(A.embed), "s"(A.token), "s"(A.hres_out), "s"(A.rms_w), "s"(A.K));
    if (PRO == OB_P_SWIGLU) asm volatile("" :: "s"(A.u_gate), "s"(A.u_up), "s"(A.st_gate), "s"(A.st_up), "s"(A.K));
    if (PRO == OB_P_PLAIN) asm volatile("" :: "s"(A.xin), "s"(A.K));
    // WGP: the workgroup's projection descriptor by uniform selects over all three (A.p[pw] as a dynamic index is a second,
    // dependent scalar-load round trip at the head of every launch)
    ObProj PW = A.p[0];
    if (WGP) {
        const ObProj &P1 = A.p[1], &P2 = A.p[2];
        asm volatile("" :: "s"(PW.w), "s"(PW.h), "s"(PW.g), "s"(PW.u), "s"(PW.st), "s"(PW.N), "s"(PW.ldw),
                     "s"(P1.w), "s"(P1.h), "s"(P1.g), "s"(P1.u), "s"(P1.st), "s"(P1.N), "s"(P1.ldw),
                     "s"(P2.w), "s"(P2.h), "s"(P2.g), "s"(P2.u), "s"(P2.st), "s"(P2.N), "s"(P2.ldw));
#define OB_PSEL(f) PW.f = pw == 0 ? PW.f : (pw == 1 ? P1.f : P2.f)
        OB_PSEL(w); OB_PSEL(h); OB_PSEL(g); OB_PSEL(u); OB_PSEL(st); OB_PSEL(N); OB_PSEL(K); OB_PSEL(ldw);
#undef OB_PSEL
    }
    const ObProj PP[3] = {PW, A.p[NPROJ > 1 ? 1 : 0], A.p[NPROJ > 2 ? 2 : 0]};
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef OB_PROFILE_ABLATE
    if (A.ablate == 4) return;              // launch floor
#endif
#ifdef OB_PROFILE_STAMPS
    // phase timestamps stay in registers and are written once at the end (a store per stamp would sit
    // in the same vmcnt queue as the loads being measured)
    unsigned long long stamp_[16] = {};
#define OB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp_[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define OB_STAMP_FLUSH() do { if (A.dbg && (threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) A.dbg[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + i_] = stamp_[i_]; } } while (0)
#else
#define OB_STAMP(i) do { } while (0)
#define OB_STAMP_FLUSH() do { } while (0)
#endif
    OB_STAMP(0);
    const int K = A.K;
    const int Kpad = (K + 511) & ~511;
    _Float16 *lds_a = reinterpret_cast<_Float16 *>(smem);
    char *lds_q = smem;
    // integer path: the digit image covers ALL KV * 8 chunks of a wave row (chunks beyond K hold zero digits, written by the
    // wave that would own them), so the MFMA phase is one straight-line block with no per-chunk guard
    constexpr int KQ = ob_dec_kq(KV);
    float *lds_red = reinterpret_cast<float *>(smem + (MATH == 1 ? (size_t)NPROJ * KQ * 4 : (size_t)NPROJ * Kpad * 2));
    // i8: lds_red holds [MT][8 waves][16 rows][4 digits] scaled fp32 partials
    constexpr int RED_OFF = MATH == 1 ? ob_dec_red_off(MT) : MT * OB_DEC_WAVES * 16;
    float *red = lds_red + RED_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gq = lane >> 4;
    const int G = wg_count;
    const int nchunks = Kpad >> 9;
    const int per_tile = (nchunks - wave + OB_DEC_WAVES - 1) / OB_DEC_WAVES;   // this wave's chunks per tile

    // slot -> tile
    int trow[MT];
    bool tval[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int ti = (j / NPROJ) * G + wg_local;
        const int ntile = (PP[j % NPROJ].N + 15) >> 4;
        tval[j] = ti < ntile;
        trow[j] = (tval[j] ? ti : 0) << 4;
    }

    // ---- 1. issue every global load (no use of any loaded value in this section) ----------------
    // 1a. prologue vectors (raw halves; indices clamped, masked later)
    bool valid[KV];
    int vbase[KV];
    ob_half8 v0[KV], v1[KV], v2[KV], hp[NPROJ][KV], vb[KV];
    _Float16 c0h = (_Float16)0, c1h = (_Float16)0;
    // the LayerNorm partials first (requests return in order and the statistics are what the prologue needs first), the
    // consumers' input_factor rows last (needed after the whole elementwise chain)
    ObTileStats<KV> ts0;
    ObTileSlot tg, tu;
    if (PST) {
        if (PRO == OB_P_SWIGLU) { ob_tiles_slot_load<KV>(tg, A.st_gate, wave, lane); ob_tiles_slot_load<KV>(tu, A.st_up, wave, lane); }
        if (PRO == OB_P_RES_LN_RMS) ob_tiles_load<KV>(ts0, A.st_prev, lane);
    } else {
        if (PRO == OB_P_SWIGLU) { c0h = A.u_gate[0]; c1h = A.u_up[0]; }
        if (PRO == OB_P_RES_LN_RMS) c0h = A.u_prev[0];
    }
#pragma unroll
    for (int v = 0; v < KV; ++v) {
        const int base = SD ? (v * OB_DEC_WAVES + wave) * 512 + (lane >> 2) * 32 + (lane & 3) * 2
                            : (v * OB_DEC_THREADS + tid) * 8;     // = chunk (v * 8 + wave), elements 8 * lane .. + 7
        valid[v] = base < K;                    // K % 32 == 0: a thread's 8 elements are all in or all out
        vbase[v] = valid[v] ? base : (SD ? (lane & 3) * 2 : 0);
        if (PRO == OB_P_PLAIN) {
            v0[v] = ob_ld8<SD>(A.xin + vbase[v]);
        } else if (PRO == OB_P_SWIGLU) {
            v0[v] = ob_ld8<SD>(A.u_gate + vbase[v]);
            v1[v] = ob_ld8<SD>(A.u_up + vbase[v]);
        } else {
            if (PRO == OB_P_RES_LN_RMS) {
                v0[v] = ob_ld8<SD>(A.u_prev + vbase[v]);
                v1[v] = ob_ld8<SD>(A.hres_in + vbase[v]);
                if (BIAS) vb[v] = ob_ld8<SD>(A.bias_prev + vbase[v]);
            }
            v2[v] = ob_ld8<SD>(A.rms_w + vbase[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < KV; ++v) {
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) hp[p][v] = ob_ld8<SD>(PP[p].h + vbase[v]);
    }
    // 1b. embedding row: the one dependent load (token id first), once per token
    if (PRO == OB_P_EMBED_RMS) {
        const _Float16 *row = A.embed + (int64_t)(*A.token) * K;
#pragma unroll
        for (int v = 0; v < KV; ++v) v1[v] = ob_ld8<SD>(row + vbase[v]);
    }
    // 1c. epilogue scale g of the output row this thread will finalise: thread (slot j, row r).
    //     Branch-free: the slot's tile / projection are SELECTED and the load is unconditional.  (A
    //     load inside "if (j == jo)" made the compiler branch around one load per slot, each preceded
    //     by s_waitcnt vmcnt(0) for the write-after-write on its destination register: every wave
    //     drained the whole prologue-vector fetch before it could request its first weight.)
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
    const _Float16 g_h = ZOUT ? (_Float16)1 : g_ptr[n_out];
    float *st_out = (tval_o && !ZOUT) ? st_sel : nullptr;
    const int tile_out = trow_o >> 4;
    __builtin_amdgcn_sched_barrier(0);
    // 1d. packed weights: items (slot j, chunk wave + 8*ci); out-of-range items re-read a valid one.
    //     ROLLING ISSUE.  With every CU streaming, one 16-byte load per lane and wave (8 KB per CU,
    //     2 MB per chip) is already the bandwidth-delay product of the HBM pipe: a wave that issues
    //     more than ~2 of them at once is parked in the issue of the third until the first returns
    //     (tools/burst_probe: 6 loads at once = 6 loads one after the other = ~1000 cycles each), and
    //     while it is parked it cannot run its share of the prologue.  So the loads are dealt out in
    //     use order: OB_ISSUE0 before the prologue, one more after each prologue stage (they return
    //     underneath the next stage), the rest inside the MFMA phase OB_ISSUE_WIN loads ahead of use.
    //     Item order = MFMA order: group g = (slot s, chunk ci) holds the NPROJ projections' loads.
#ifndef OB_ISSUE0
#define OB_ISSUE0 2
#endif
#ifndef OB_ISSUE_STEP
#define OB_ISSUE_STEP 1
#endif
#ifndef OB_ISSUE_WIN
#define OB_ISSUE_WIN 2
#endif
#ifndef OB_HEAD_BARRIER
#define OB_HEAD_BARRIER 1
#endif
#if OB_HEAD_BARRIER
    // Round 4: every wave's prologue-vector requests enter the CU's memory pipeline before any wave's weight stream.
    // Waves of a workgroup start hundreds of cycles apart and the vector L1 returns in request order: a late wave's
    // 1 KB vector slices queued behind the early waves' HBM misses, and the first workgroup barrier then waited for that
    // wave (tools/phase_probe.py: statistics ready 1700 median / 3700 max).  1055 -> 1085 tok/s; -DOB_HEAD_BARRIER=0: A/B.
    __builtin_amdgcn_s_barrier();
#endif
    ob_u32x4 wreg[MT][KV];
    constexpr int NITEM = MT * KV, NG = MS * KV;
#ifndef OB_ISSUE0_SWIGLU
#define OB_ISSUE0_SWIGLU OB_ISSUE0
#endif
    constexpr int I0 = PRO == OB_P_SWIGLU ? OB_ISSUE0_SWIGLU : OB_ISSUE0;     // (A/B: fewer weight requests ahead of the long SwiGLU prologue)
    constexpr int C0 = NITEM < I0 ? NITEM : I0;
    constexpr int C1 = NITEM < C0 + OB_ISSUE_STEP ? NITEM : C0 + OB_ISSUE_STEP;
    constexpr int C2 = NITEM < C1 + OB_ISSUE_STEP ? NITEM : C1 + OB_ISSUE_STEP;
    constexpr int C3 = NITEM < C2 + OB_ISSUE_STEP ? NITEM : C2 + OB_ISSUE_STEP;
    constexpr int C4 = NITEM < C3 + OB_ISSUE_STEP ? NITEM : C3 + OB_ISSUE_STEP;
    auto load_items = [&](int first, int last) {
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            if (it >= first && it < last) {
                const int g = it / NPROJ, p = it % NPROJ, ci = g % KV, j = (g / KV) * NPROJ + p;
#ifdef OB_PROFILE_ABLATE
                if (A.ablate == 2 || A.ablate == 3) { wreg[j][ci] = (ob_u32x4){0x12345678u + lane, 0x9abcdef0u, 0x0f1e2d3cu, 0x55aa55aau}; continue; }
#endif
                wreg[j][ci] = ob_dec_load_w<ALIGNED>(PP[p].w, PP[p].N, K, PP[p].ldw, trow[j], min(wave + ci * OB_DEC_WAVES, nchunks - 1), lane);
            }
        }
    };
    load_items(0, C0);
    __builtin_amdgcn_sched_barrier(0);      // nothing above may sink below, no use may rise above
    OB_STAMP(1);
#define OB_ISSUE(a, b) do { __builtin_amdgcn_sched_barrier(0); load_items(a, b); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- 2. prologue arithmetic in the reference's op order; fp16 tensor ops are native packed
    //         fp16 instructions (contraction off: every op rounds once, as torch does) -------------
    ob_half8 xh[KV];
    if (PRO == OB_P_PLAIN) {
#pragma unroll
        for (int v = 0; v < KV; ++v) xh[v] = v0[v];
        OB_ISSUE(C0, C2);
    } else if (PRO == OB_P_SWIGLU) {
        float mg, rg, mu, ru;
        if (PST) {
            const ob_float4 pg = ob_tiles_slot_partial<KV>(tg, K, wave, lane), pu = ob_tiles_slot_partial<KV>(tu, K, wave, lane);
            if (lane == 0) {
                *reinterpret_cast<ob_float4 *>(red + wave * 8) = pg;
                *reinterpret_cast<ob_float4 *>(red + wave * 8 + 4) = pu;
            }
            __syncthreads();
            OB_ISSUE(C0, C1);
            ob_tiles_slot_merge<OB_DEC_WAVES>(red, 0, K, A.ln_eps, mg, rg);
            ob_tiles_slot_merge<OB_DEC_WAVES>(red, 1, K, A.ln_eps, mu, ru);
        } else {
            const float c0 = (float)c0h, c1 = (float)c1h;
            ob_float2 sg2 = {0.f, 0.f}, qg2 = {0.f, 0.f}, su2 = {0.f, 0.f}, qu2 = {0.f, 0.f};
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                if (valid[v]) { ob_stats8(v0[v], c0, sg2, qg2); ob_stats8(v1[v], c1, su2, qu2); }
            }
            float s[4] = {sg2[0] + sg2[1], qg2[0] + qg2[1], su2[0] + su2[1], qu2[0] + qu2[1]};
            ob_block_sum_n<4, OB_DEC_WAVES>(s, red);
            OB_ISSUE(C0, C1);
            ob_ln_stats(s[0], s[1], c0, K, A.ln_eps, mg, rg);
            ob_ln_stats(s[2], s[3], c1, K, A.ln_eps, mu, ru);
        }
        OB_STAMP(2);
        const float ng = -mg * rg, nu = -mu * ru;
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            ob_half8 sg, up;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const _Float16 gh = ob_ln_apply_h(v0[v][i], rg, ng);                     // LayerNorm(gate) -> fp16
                up[i] = ob_ln_apply_h(v1[v][i], ru, nu);                                // LayerNorm(up)   -> fp16
                // silu -> fp16: gate * 1 / (1 + 2^(-gate * log2 e)); the fp16 operands ride in the fma_mix forms
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf((float)gh, -1.44269504088896341f, 0.0f));
                sg[i] = (_Float16)__builtin_fmaf((float)gh, __builtin_amdgcn_rcpf(1.0f + e), 0.0f);
            }
            xh[v] = sg * up;                                    // act_fn(gate) * up, modeling_bitllama.py:257
        }
        OB_ISSUE(C1, C2);
    } else {
        ob_half8 hv[KV];
        if (PRO == OB_P_RES_LN_RMS) {
            float mean, rstd;
            if (PST) {
                ob_tiles_combine<KV>(ts0, K, A.ln_eps, lane, mean, rstd);
                OB_ISSUE(C0, C1);
            } else {
                const float c0 = (float)c0h;
                ob_float2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                for (int v = 0; v < KV; ++v) {
                    if (valid[v]) ob_stats8(v0[v], c0, s2, q2);
                }
                float s[2] = {s2[0] + s2[1], q2[0] + q2[1]};
                ob_block_sum_n<2, OB_DEC_WAVES>(s, red);
                OB_ISSUE(C0, C1);
                ob_ln_stats(s[0], s[1], c0, K, A.ln_eps, mean, rstd);
            }
            OB_STAMP(2);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                ob_half8 ln;
#pragma unroll
                for (int i = 0; i < 8; ++i) ln[i] = ob_ln_apply_h(v0[v][i], rstd, nmr);
                if (BIAS) ln = ln + vb[v];          // output += bias, bitnet.py:119-120 (one fp16 rounding)
                hv[v] = v1[v] + ln;                 // residual + hidden_states, modeling_bitllama.py:912,918
            }
        } else {
#pragma unroll
            for (int v = 0; v < KV; ++v) hv[v] = v1[v];
            OB_ISSUE(C0, C1);
        }
        // RMSNorm (modeling_bitllama.py:76-81): fp32 variance, x * rsqrt -> fp16, weight * that -> fp16
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
        ob_block_sum_n<1, OB_DEC_WAVES>(ss, red + 64);
        OB_STAMP(3);
        OB_ISSUE(C1, C2);
        const float rs = __builtin_amdgcn_rsqf(ss[0] * __builtin_amdgcn_rcpf((float)K) + A.rms_eps);
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            ob_half8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (_Float16)__builtin_fmaf((float)hv[v][i], rs, 0.0f);   // fp16(h * rsqrt)
            xh[v] = v2[v] * t;
            if (blockIdx.x == 0 && A.hres_out && valid[v]) ob_st8<SD>(A.hres_out + vbase[v], hv[v]);
        }
    }
    OB_STAMP(4);

    if (MATH == 0) {
        // a_p = fp16(x * h_p)  (bitnet.py:113), zero padding up to Kpad
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) {
            _Float16 *dst = lds_a + (size_t)p * Kpad;
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                const int base = (v * OB_DEC_THREADS + tid) * 8;
                if (base < Kpad) {
                    ob_half8 o = xh[v] * hp[p][v];
                    if (!valid[v]) o = (ob_half8)(_Float16)0;
                    *reinterpret_cast<ob_half8 *>(dst + base) = o;
                }
            }
        }
        // no barrier: chunk (wave + 8*ci) is exactly the elements this wave's lanes hold for vector
        // ci, so every wave reads back only what it wrote itself (LDS ops of one wave are in order)
        OB_STAMP(6);
        OB_ISSUE(C2, NITEM);

        // ---- 3. MFMA ---------------------------------------------------------------------------
        ob_float4 acc[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            acc[j] = (ob_float4){0.f, 0.f, 0.f, 0.f};
            const _Float16 *ap = lds_a + (size_t)(j % NPROJ) * Kpad;
#pragma unroll
            for (int ci = 0; ci < KV; ++ci) {
                if (tval[j] && ci < per_tile) {
#ifdef OB_PROFILE_ABLATE
                    if (A.ablate == 1 || A.ablate == 3) {
                        acc[j][0] += __uint_as_float((wreg[j][ci][0] ^ wreg[j][ci][1] ^ wreg[j][ci][2] ^ wreg[j][ci][3]) & 0x3fffffffu);
                        continue;
                    }
#endif
                    ob_dec_chunk(wreg[j][ci], ap + (size_t)(wave + ci * OB_DEC_WAVES) * 512, gq, acc[j]);
                }
            }
        }
        OB_STAMP(8);
        // ---- 4. cross-wave reduction: column 0 (lanes 0,16,32,48 hold rows 4*gq..4*gq+3)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            if ((lane & 15) == 0) {
                float *dst = lds_red + ((j * OB_DEC_WAVES + wave) << 4) + 4 * gq;
                dst[0] = acc[j][0]; dst[1] = acc[j][1]; dst[2] = acc[j][2]; dst[3] = acc[j][3];
            }
        }
        __syncthreads();
        float uval = 0.f;
        if (fin) {
            const int r = tid & 15;
            float z = 0.f;
#pragma unroll
            for (int w = 0; w < OB_DEC_WAVES; ++w) z += lds_red[((jo * OB_DEC_WAVES + w) << 4) + r];
            // z -> fp16 (bitnet.py:115), * g -> fp16 (:116)
            if (ZOUT) {
                reinterpret_cast<float *>(u_out)[n_out] = z;
            } else {
                const _Float16 uh = (_Float16)(ob_round_h(z) * (float)g_h);
                u_out[n_out] = uh;
                uval = (float)uh;
            }
        }
        // per-tile LayerNorm partials for the consumer kernels: the 16 rows of a tile are one DPP row
        if (!ZOUT && tid < MT * 16) {
            const float sm = ob_row16_sum(fin ? uval : 0.f);     // st_out is only given when N % 16 == 0: full tiles
            const float dv = fin ? uval - sm * 0.0625f : 0.f;
            const float m2 = ob_row16_sum(dv * dv);
            if (st_out && (tid & 15) == 0) *reinterpret_cast<ob_float2 *>(st_out + 2 * tile_out) = (ob_float2){sm, m2};
        }
        OB_STAMP(10);
    } else {
        // ---- integer path ----------------------------------------------------------------------
        // 2b. a_p = fp16(x * h_p); everything from here to the MFMAs is wave-local: chunk
        //     (wave + 8*ci) is exactly the elements this wave's lanes hold for vector ci, so each wave
        //     quantises its own elements relative to ITS largest one (exponent e per wave and
        //     projection), writes the digits, reads them back and multiplies -- no workgroup barrier,
        //     and one wave's VALU work overlaps another wave's MFMAs on the same SIMD.
        ob_half8 ah[NPROJ][KV];
        int e_w[NPROJ];
        bool nonfinite[NPROJ];
#pragma unroll
        for (int v = 0; v < KV; ++v)
            if (!valid[v]) xh[v] = (ob_half8)(_Float16)0;          // zero padding up to Kpad
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) {
            ob_u16x2 mx = {0, 0};
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                ah[p][v] = SD ? xh[v] * hp[p][v] : ob_quad_transpose(xh[v] * hp[p][v], lane);    // -> strided ownership (see ob_ld8)
                const ob_u32x4 bits = __builtin_bit_cast(ob_u32x4, ah[p][v]);
#pragma unroll
                for (int d = 0; d < 4; ++d)             // |a| as fp16 bit patterns order like unsigned integers
                    mx = __builtin_elementwise_max(mx, __builtin_bit_cast(ob_u16x2, bits[d] & 0x7fff7fffu));
            }
            const uint32_t m = ob_wave_max_u32(max((uint32_t)mx[0], (uint32_t)mx[1]));
            e_w[p] = (int)max(m >> 10, 1u) - 15;        // |a| < 2^(e+1) for every element of this wave
            nonfinite[p] = m >= 0x7c00u;                // an Inf / NaN activation: fixed point cannot carry it
        }
        OB_STAMP(5);
        OB_ISSUE(C2, C3);
        // this lane's two bit positions j = 2jp + s: compensation 2^(7-j) (j < 7) or -1 (j = 7) folded
        // into the quantisation scale, v_j = 2^j / -128 as the byte of the S dot product
        const int jp = lane & 3;
        const float cj0 = __uint_as_float((uint32_t)(127 + 7 - 2 * jp) << 23);
        const float cj1 = jp < 3 ? __uint_as_float((uint32_t)(127 + 6 - 2 * jp) << 23) : -1.0f;
        const int vj0 = (int)(0x01010101u << (2 * jp)), vj1 = (int)(0x01010101u << (2 * jp + 1));
        float inv_scale[NPROJ];
        int sdig[NPROJ];
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) {
            const int e = e_w[p];
            const float scale = __uint_as_float((uint32_t)(22 - e + 127) << 23);          // 2^(22-e)
            inv_scale[p] = __uint_as_float((uint32_t)(e - 29 + 127) << 23);               // 2^-(22-e+7)
            const float sc0 = scale * cj0, sc1 = scale * cj1;
            int D[4] = {0, 0, 0, 0};
            char *dst = lds_q + (size_t)p * KQ * 4 + (lane >> 2) * 128 + jp * 8;
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                uint32_t T[2][4];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint32_t W[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // fma(a, s, 0) = a * s exactly; written as an fma so that the fp16 -> fp32
                        // conversion rides in the same instruction (v_fma_mix_f32)
                        const float f = __builtin_fmaf((float)ah[p][v][2 * i + s2], s2 ? sc1 : sc0, 0.0f);
                        int m;
                        asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(m) : "v"(f));      // floor(f + 0.5)
                        W[i] = ((uint32_t)m + 0x00808080u) ^ 0x00808080u;         // 4 signed digits [d0 d1 d2 d3]
                    }
                    // 4x4 byte transpose: T[c] = digit c of elements i = 0..3
                    const uint32_t u0 = __builtin_amdgcn_perm(W[1], W[0], 0x05010400u), u1 = __builtin_amdgcn_perm(W[1], W[0], 0x07030602u);
                    const uint32_t w0 = __builtin_amdgcn_perm(W[3], W[2], 0x05010400u), w1 = __builtin_amdgcn_perm(W[3], W[2], 0x07030602u);
                    T[s2][0] = __builtin_amdgcn_perm(w0, u0, 0x05040100u);
                    T[s2][1] = __builtin_amdgcn_perm(w0, u0, 0x07060302u);
                    T[s2][2] = __builtin_amdgcn_perm(w1, u1, 0x05040100u);
                    T[s2][3] = __builtin_amdgcn_perm(w1, u1, 0x07060302u);
                    if (!SMF) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) D[c] = __builtin_amdgcn_sdot4((int)T[s2][c], s2 ? vj1 : vj0, D[c], false);
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)         // (a chunk beyond K: xh is zero there, so are its digits)
                    *reinterpret_cast<ob_u32x2 *>(dst + (size_t)(v * OB_DEC_WAVES + wave) * 2048 + c * 32) = (ob_u32x2){T[0][c], T[1][c]};
            }
            // S of this wave, exact, digit (lane & 3) in every lane.  OB_SMFMA (default): S comes out of the matrix
            // pipe itself -- eight more MFMAs per chunk and projection whose A operand is the all-ones word's masks
            // (compile-time constants: no v_and), i.e. "every sign bit set": 128 * S_c lands in the accumulator layout,
            // in the lanes that hold 128 * B_c.  The pipe has the room (48 of ~190 issue slots per wave); the VALU loses
            // 8 v_dot4 per 8 elements and the 12-step transposing cross-lane reduction per projection.
            // OB_SMFMA=0: v_dot4 per element + a transposing reduction (4 values -> 2 -> 1 across lane ^ 2, lane ^ 1),
            // rotate-adds inside the 16-lane rows, then the two gfx950 row swaps
            if (!SMF) {
            const bool hi2 = lane & 2, hi1 = lane & 1;
            int k0 = hi2 ? D[2] : D[0], k1 = hi2 ? D[3] : D[1];
            const int s0 = hi2 ? D[0] : D[2], s1 = hi2 ? D[1] : D[3];
            k0 += __builtin_amdgcn_update_dpp(0, s0, 0x4E, 0xF, 0xF, false);    // lane ^ 2
            k1 += __builtin_amdgcn_update_dpp(0, s1, 0x4E, 0xF, 0xF, false);
            int t = hi1 ? k1 : k0;
            const int sd = hi1 ? k0 : k1;
            t += __builtin_amdgcn_update_dpp(0, sd, 0xB1, 0xF, 0xF, false);     // lane ^ 1
            t += __builtin_amdgcn_update_dpp(0, t, 0x124, 0xF, 0xF, false);     // row_ror:4
            t += __builtin_amdgcn_update_dpp(0, t, 0x128, 0xF, 0xF, false);     // row_ror:8
            auto s16 = __builtin_amdgcn_permlane16_swap((uint32_t)t, (uint32_t)t, false, false);
            t = (int)(s16[0] + s16[1]);
            auto s32 = __builtin_amdgcn_permlane32_swap((uint32_t)t, (uint32_t)t, false, false);
            sdig[p] = (int)(s32[0] + s32[1]);
            }
        }
        OB_STAMP(6);
        OB_ISSUE(C3, C4);

        // 3. MFMA, group by group in issue order: for every (word q, half jh) of group (s, ci) ONE
        //    activation read per projection, then one MFMA per projection -- consecutive instructions
        //    hit different accumulators.  Before a group is multiplied the loads OB_ISSUE_WIN items
        //    ahead of it are issued (the wave has nothing else left to do, so parking in that issue costs
        //    nothing).  Invalid slots re-run a valid tile (never stored).
        const int cpc = lane & 3;
        ob_i32x4 acc[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[j] = (ob_i32x4){0, 0, 0, 0};
        const char *bq = lds_q + (size_t)(gq * 16 + cpc) * 32;
        // B operands (this wave's digit planes of one 512-weight chunk and projection: 8 x 16 bytes per lane) are
        // read in BULK, 8 * NPROJ ds_read_b128 back to back behind one wait, and kept in registers: all MS slots
        // of a projection multiply the same activations (with KV = 1 the chunk is the same for every group: one
        // bulk read per launch), and the LDS serves a stream of independent reads at its full rate where it served
        // read -> wait -> MFMA chains at a fraction of it (MI355X_MICROARCH.md, LDS: >= 16 DS operations per wait;
        // the gate|up launch issued 48 dependent reads per wave for 16 distinct operands).  OB_BCACHE=0: the old form.
#ifndef OB_BCACHE
#define OB_BCACHE 1
#endif
        ob_i32x4 bc[NPROJ][8];
        ob_i32x4 acc_s[NPROJ];
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) acc_s[p] = (ob_i32x4){0, 0, 0, 0};
        const ob_i32x4 ones_lo = {0x01010101, 0x02020202, 0x04040404, 0x08080808};
        const ob_i32x4 ones_hi = {0x10101010, 0x20202020, 0x40404040, (int)0x80808080u};
        // IL accumulators are interleaved per MFMA step: the projections of a slot (per-slot form) or two slots of the
        // workgroup's projection sharing ONE B operand (WGP) -- consecutive MFMAs never chain on one accumulator.
        constexpr int IL = WGP ? (MS >= 2 ? 2 : 1) : NPROJ;
        constexpr int NGRP = WGP ? ((MS + IL - 1) / IL) * KV : NG;
        // (No per-chunk guard: a wave row's chunks beyond K multiply zero digits -- the guard made every group its own basic
        //  block, the accumulators travelled through phi copies at the merges, and with two guarded copies of the phase the
        //  waitcnt pass fell back to ONE vmcnt(0) in front of the first MFMA: the whole weight stream had to land first.)
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int s = g / KV, ci = g % KV;          // s: slot (per-slot form) or slot pair (WGP)
            {
                // items consumed up to and including group g (WGP: all chunks of the slot pair at once)
                const int need_prev = g == 0 ? 0 : (WGP ? ((g - 1) / KV + 1) * IL * KV : g * NPROJ);
                const int need = WGP ? (s + 1) * IL * KV : (g + 1) * NPROJ;
                const int prev = g == 0 ? C4 : ((need_prev + OB_ISSUE_WIN) > C4 ? (need_prev + OB_ISSUE_WIN) : C4);
                const int want = (need + OB_ISSUE_WIN) > C4 ? (need + OB_ISSUE_WIN) : C4;
                if (prev < NITEM && want > prev) OB_ISSUE(prev < NITEM ? prev : NITEM, want < NITEM ? want : NITEM);
            }
            {
                const int ch = wave + ci * OB_DEC_WAVES;
                constexpr int NB = WGP ? 1 : NPROJ;      // distinct B operands per step
                if (OB_BCACHE && (KV > 1 || s == 0)) {
#pragma unroll
                    for (int p = 0; p < NB; ++p)
#pragma unroll
                        for (int qj = 0; qj < 8; ++qj)
                            bc[p][qj] = *reinterpret_cast<const ob_i32x4 *>(bq + (size_t)p * KQ * 4 + (size_t)ch * 2048 + (qj >> 1) * 128 + (qj & 1) * 16);
                    __builtin_amdgcn_sched_barrier(0);          // the reads stay one burst (hipcc otherwise sinks each next to its MFMA)
                }
                // lane l of the step: slot jl(l), B operand bl(l); WGP: the last pair of an odd MS has one lane
                auto jl = [&](int l) { return WGP ? s * IL + l : s * NPROJ + l; };
                const int nl = WGP ? ((s + 1) * IL <= MS ? IL : MS - s * IL) : NPROJ;
                const bool smf_here = SMF && s == 0;
                // steps t = (q, jh) of the chunk, software-pipelined by one: the A operands (w & mask, 4 v_and per MFMA) of
                // step t + 1 are formed BEFORE the MFMAs of step t are issued, in a second register set -- a v_and
                // result feeding the very next instruction costs MFMA-hazard wait states and chains every MFMA behind
                // its own operand preparation (hipcc reused ONE register quad for all 48 operands: 35 s_nop per wave)
                auto masks = [&](int t, int l) -> ob_i32x4 {
                    const uint32_t w = wreg[jl(l)][ci][t >> 1];
                    ob_i32x4 av;
#pragma unroll
                    for (int v = 0; v < 4; ++v) av[v] = (int)(w & (0x01010101u << (4 * (t & 1) + v)));
                    return av;
                };
                ob_i32x4 avn[IL];
#pragma unroll
                for (int l = 0; l < IL; ++l) if (l < nl) avn[l] = masks(0, l);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    ob_i32x4 avc[IL], bv[NB];
#pragma unroll
                    for (int l = 0; l < IL; ++l) if (l < nl) avc[l] = avn[l];
#pragma unroll
                    for (int p = 0; p < NB; ++p)
                        bv[p] = OB_BCACHE ? bc[p][t]
                                          : *reinterpret_cast<const ob_i32x4 *>(bq + (size_t)p * KQ * 4 + (size_t)ch * 2048 + (t >> 1) * 128 + (t & 1) * 16);
                    if (t < 7) {
#pragma unroll
                        for (int l = 0; l < IL; ++l) if (l < nl) avn[l] = masks(t + 1, l);
                    }
#pragma unroll
                    for (int l = 0; l < IL; ++l) {
                        if (l < nl) {
                            const int j = jl(l);
                            acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(avc[l], bv[WGP ? 0 : l], acc[j], 0, 0, 0);
                            if (smf_here && (!WGP || l == 0))  // 128 * S of this chunk: the all-ones word through the same step
                                acc_s[WGP ? 0 : l] = __builtin_amdgcn_mfma_i32_16x16x64_i8((t & 1) ? ones_hi : ones_lo, bv[WGP ? 0 : l], acc_s[WGP ? 0 : l], 0, 0, 0);
                        }
                    }
#if OB_BCACHE
                    // (the builtin wants literal counts: nl and the MFMA count are constants only after unrolling)
                    const int nm = nl + (smf_here ? (WGP ? 1 : nl) : 0);
                    if (t < 7) {                                                                            // VALU: next step's masks
                        if (nl == 1) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        else if (nl == 2) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                    }
                    if (nm == 1) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         // MFMA: this step
                    else if (nm == 2) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    else if (nm == 3) __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    else if (nm == 4) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#endif
#ifdef OB_PROFILE_STAMPS
                    if (g == 0 && t == 0) OB_STAMP(7);
#endif
                }
            }
        }
        OB_STAMP(8);
        // 4. per wave, row and digit c: (S_c - 2 B_c) is exact in int32; one conversion to fp32, scaled
        //    by 2^(8c) / (128 * 2^(22-e)) (powers of two: exact), then a fixed-order fp32 sum over the
        //    4 digits and 8 waves in the finishing thread -- deterministic, error <= a few 2^-24
        if ((lane & 15) < 4) {
            const float dscale = __uint_as_float((uint32_t)(127 + 8 * (lane & 3)) << 23);
            // (S - 2 B) * f as fma((float)B, -2 f, (float)S * f): f is a power of two and |S|, |B|, |S - 2 B| < 2^24,
            // so every step is exact and the value is the same -- two instructions per row instead of four
            float fa[NPROJ], fb[NPROJ];
#pragma unroll
            for (int p = 0; p < NPROJ; ++p) {
                // a non-finite activation makes the whole output row NaN, as it does in the reference's GEMM
                const float f = nonfinite[p] ? __builtin_nanf("") : dscale * inv_scale[p];
                fa[p] = (float)(SMF ? acc_s[p][0] : sdig[p]) * f;     // (all 16 rows of the ones operand are the same row)
                fb[p] = -2.0f * f;
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float *dst = lds_red + (((j * OB_DEC_WAVES + wave) * 16 + 4 * gq) << 2) + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[4 * r] = __builtin_fmaf((float)acc[j][r], fb[j % NPROJ], fa[j % NPROJ]);
            }
        }
        __syncthreads();
        OB_STAMP(9);
        float uval = 0.f;
        if (fin) {
            const int r = tid & 15;
            // fixed order: (sum over waves of digit 0 + digit 2) + (sum over waves of digit 1 + digit 3), as packed
            // fp32 adds on the register pairs the 16-byte reads deliver (hipcc turned the scalar form
            // ((t0 + t1) + (t2 + t3)) into packed adds behind 30 register moves)
            ob_float2 z2 = {0.f, 0.f};
#pragma unroll
            for (int w = 0; w < OB_DEC_WAVES; ++w) {
                const ob_float4 t = *reinterpret_cast<const ob_float4 *>(lds_red + (((jo * OB_DEC_WAVES + w) * 16 + r) << 2));
                const ob_float2 lo = {t[0], t[1]}, hi = {t[2], t[3]};
                z2 += lo + hi;
            }
            const float z = z2[0] + z2[1];
            if (ZOUT) {
                reinterpret_cast<float *>(u_out)[n_out] = z;             // fp32 partial sum of this K slice (rounded by the consumer)
            } else {
                const _Float16 uh = (_Float16)(ob_round_h(z) * (float)g_h);   // fp16(z) (bitnet.py:115), * g -> fp16 (:116)
                u_out[n_out] = uh;
                uval = (float)uh;
            }
        }
        // per-tile LayerNorm partials for the consumer kernels: the 16 rows of a tile are one DPP row
        if (!ZOUT && tid < MT * 16) {
            const float sm = ob_row16_sum(fin ? uval : 0.f);     // st_out is only given when N % 16 == 0: full tiles
            const float dv = fin ? uval - sm * 0.0625f : 0.f;
            const float m2 = ob_row16_sum(dv * dv);
            if (st_out && (tid & 15) == 0) *reinterpret_cast<ob_float2 *>(st_out + 2 * tile_out) = (ob_float2){sm, m2};
        }
        OB_STAMP(10);
    }
    if (PRO == OB_P_EMBED_RMS) {
        if (blockIdx.x == 0 && A.rope_out && tid < 2 * A.rope_D) {
            const int ps = min(max(*A.rope_pos, 0), A.rope_max - 1);
            const int d = tid < A.rope_D ? tid : tid - A.rope_D;
            A.rope_out[tid] = (tid < A.rope_D ? A.rope_cos : A.rope_sin)[(int64_t)ps * A.rope_D + d];
        }
    }
    OB_STAMP_FLUSH();
#undef OB_STAMP
#undef OB_STAMP_FLUSH
#undef OB_ISSUE
}

// ---------------------------------------------------------------------------------------------
// Attention for one new token (modeling_bitllama.py:522-563), one 256-thread workgroup per head.
// ---------------------------------------------------------------------------------------------
struct ObAttnArgs {
    const _Float16 *u_q, *u_k, *u_v;     // pre-LayerNorm projections [H*D], [Hkv*D], [Hkv*D]
    const _Float16 *cos, *sin;           // rope tables [max_pos, D] (fp16, as the reference caches them)
    _Float16 *kcache, *vcache;           // [Hkv, max_len, D]
    _Float16 *out;                       // [H*D]
    const int *pos;                      // device [slots]: position of the new token (= tokens already cached); < 0 = idle slot
    const float *st_q, *st_k, *st_v;     // PST kernels: per-tile (sum, M2) partials of u_q / u_k / u_v from the q|k|v GEMV
    int H, Hkv, D, max_len;
    float ln_eps;
    long long slot_stride;               // elements between the caches of consecutive slots (blockIdx.y); rows of
                                         // u_q / u_k / u_v / out are consecutive per slot
    const _Float16 *rope_cur;            // optional [2 * D]: cos[pos] | sin[pos] copied by the step's first launch (no pos -> table chase here)
    const _Float16 *h_next;              // optional [H * D]: out <- fp16(out * h_next), o_proj's input scaling (bitnet.py:113)
                                         // for a consumer that takes pre-scaled rows (batched step, ob_skinny3.h)
    const _Float16 *b_q, *b_k, *b_v;     // BIAS kernels (config.attention_bias, modeling_bitllama.py:451-453): q / k / v =
                                         // fp16(LayerNorm(u) + b) (bitnet.py:118-120) before the rotary embedding
    // ZIN kernels (K-sharded decode step): the three rows as COMPLETE fp32 sums + weight_scale; u = fp16(fp16(z) * g)
    // (bitnet.py:115-116) is formed on the fly and the statistics are recomputed per workgroup (u_q / u_k / u_v, st_* unused)
    const float *z_q, *z_k, *z_v;
    const _Float16 *g_q, *g_k, *g_v;
};

__device__ __forceinline__ _Float16 ob_zg(const float *z, const _Float16 *g, int i) { return (_Float16)(ob_round_h(z[i]) * (float)g[i]); }
__device__ __forceinline__ ob_half8 ob_zg8(const float *z, const _Float16 *g, int base)
{
    const ob_float4 z0 = *reinterpret_cast<const ob_float4 *>(z + base), z1 = *reinterpret_cast<const ob_float4 *>(z + base + 4);
    const ob_half8 gv = *reinterpret_cast<const ob_half8 *>(g + base);
    ob_half8 u;
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] = (_Float16)(ob_round_h(z0[i]) * (float)gv[i]); u[4 + i] = (_Float16)(ob_round_h(z1[i]) * (float)gv[4 + i]); }
    return u;
}

// Thread (pg, ds) = (tid >> 4, tid & 15): position group pg (32 of them, positions pg + 32 i) and
// 8-dim slice ds of the head; a 16-lane DPP row spans the head dimension, so the q.k dots are row
// reductions and every lane of a row holds its row's scores and probabilities -- no LDS round trip
// between scores, softmax and P.V.  Five barriers: LayerNorm statistics, q/k/v of the new token,
// softmax maximum, softmax denominator, output partials.  D <= 128.
#define OB_ATTN_THREADS 512
#define OB_ATTN_WAVES (OB_ATTN_THREADS / 64)
__device__ __forceinline__ float ob_row_sum(float v)     // sum over the 16 lanes of a DPP row, in every lane
{
    v += OB_DPP_F(v, 0xB1, 0xF);
    v += OB_DPP_F(v, 0x4E, 0xF);
    v += OB_DPP_F(v, 0x141, 0xF);
    v += OB_DPP_F(v, 0x140, 0xF);
    return v;
}
__device__ __forceinline__ float ob_rows_sum(float v)    // row-uniform values: sum over the wave's 4 rows
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float ob_rows_max(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// NTH = 512 or 256 threads.  The kernel is a chain of dependent phases on 32 workgroups; a wave that
// has its SIMD to itself issues an instruction every ~4.7 cycles against ~9 when two waves share it
// (tools/issue_probe.hip), and most of a wave's instructions here do not scale with its share of the
// positions -- so the single-sequence decode step runs it with 4 waves (one per SIMD), 8 positions
// per thread and sweep; the batched step (many workgroups per CU anyway) keeps 8 waves.
// BLIND: fetch the first 128 cached positions before the position is known (single sequence: 32 workgroups,
// latency is everything).  The batched step runs heads x slots workgroups -- 64 KB of blind cache reads each
// were 64 MB per layer at 32 slots, 13 of the kernel's 16 us -- and reads the position first.
// PF (single sequence only): the launch runs one workgroup per head on a 256-CU chip -- the grid carries one extra
// workgroup per idle CU that does nothing but pull o_proj's packed rows (the next launch: 2 MB) into the L2 of the XCD
// whose workgroups will read them (ob_common.h).
// NB (round 5): positions whose K / V rows are requested before the position is known and whose scores stay in registers (the
// "blind window").  128 reads 64 KB per head workgroup whatever the context holds; a step the HOST knows to be at a position
// < 64 takes the NB = 64 instance (half the rows: the launch is bound by that fetch, not by its arithmetic).
template <bool PST, int NTH, bool BLIND = true, bool BIAS = false, bool ZIN = false, int NB = 128>
__global__ __launch_bounds__(NTH) void ob_dec_attn_kernel(const ObAttnArgs A_in, const ObPfPlan PF)
{
    static_assert(!ZIN || !PST, "ZIN recomputes the statistics from the sums: non-PST form");
    static_assert(NB == 128 || NB == 64, "blind window");
    ObAttnArgs A = A_in;
    if (BLIND) {
        // every kernel argument is requested in ONE scalar-load clause (hipcc fetches fields where they are first used:
        // four dependent s_load / s_waitcnt round trips stood between the kernel entry and the last vector request)
        asm volatile("" :: "s"(A.u_q), "s"(A.u_k), "s"(A.u_v), "s"(A.cos), "s"(A.sin), "s"(A.kcache), "s"(A.vcache), "s"(A.out),
                     "s"(A.pos), "s"(A.st_q), "s"(A.st_k), "s"(A.st_v), "s"(A.rope_cur), "s"(A.h_next), "s"(A.H), "s"(A.Hkv),
                     "s"(A.D), "s"(A.max_len), "s"(A.ln_eps));
    }
    if (BLIND && ob_prefetch_only_wg(PF, A_in.H, (int)threadIdx.x, NTH)) return;
    constexpr int NWV = NTH / 64, PG = NTH / 16, NI = NB / PG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = A.D, H = A.H, Hkv = A.Hkv;
    const int head = blockIdx.x, kvh = head / (H / Hkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // sequence slot (batched step): one row of the projections and one cache per slot
        const int slot = blockIdx.y;
        A.u_q += (int64_t)slot * H * D; A.u_k += (int64_t)slot * Hkv * D; A.u_v += (int64_t)slot * Hkv * D;
        A.out += (int64_t)slot * H * D;
        A.kcache += (int64_t)slot * A.slot_stride; A.vcache += (int64_t)slot * A.slot_stride;
        A.pos += slot;
        if (PST) {      // (batched step: one block of partials per slot)
            A.st_q += (size_t)slot * (((H * D + 4095) >> 12) * 512);
            A.st_k += (size_t)slot * (((Hkv * D + 4095) >> 12) * 512); A.st_v += (size_t)slot * (((Hkv * D + 4095) >> 12) * 512);
        }
    }
    float *red = reinterpret_cast<float *>(smem);                    // [0,96) stats, [96,104) max, [112,120) sum
    _Float16 *q_s = reinterpret_cast<_Float16 *>(smem + 512);         // [128] query (post RoPE), zero padded
    _Float16 *k_s = q_s + 128;                                       // [128] new key (post RoPE)
    _Float16 *v_s = k_s + 128;                                       // [128] new value
    float *po = reinterpret_cast<float *>(v_s + 128);                // [8 waves][128] partial outputs
    float *sc = po + NWV * 128;                            // [max_len] scores of positions >= 128

    // ---- every load of the short-context path is issued here ------------------------------------
    // The first 128 cached positions are fetched WITHOUT waiting for the position (one dependent
    // round trip less on the critical path): rows at or beyond it hold stale or never-written
    // data and are masked below (scores by select, values by branch), only max_len bounds them.
    const int NQ = H * D, NK = Hkv * D;
    const _Float16 *kbase = A.kcache + (int64_t)kvh * A.max_len * D;
    const _Float16 *vbase = A.vcache + (int64_t)kvh * A.max_len * D;
    const int ds = tid & 15, pg = tid >> 4;
    const bool dok = 8 * ds < D;
    const int dcl = dok ? 8 * ds : 0;
    ob_half8 kreg[NI], vreg[NI];
    // everything that does not depend on the position is requested before the position is read (batched step: the
    // position, then the K / V rows, then the rest were three dependent round trips)
    // PST (round 4): ROLES.  Wave 0 normalises and rotates q, wave 1 k, wave 2 v -- each combines only ITS vector's tile
    // partials (every wave combining all three and two waves doing all the LayerNorm / RoPE arithmetic was ~100
    // instructions of the chain in front of the first barrier) and a lane holds both elements of a rotate_half pair.
    const int half = D >> 1;
    const int role = __builtin_amdgcn_readfirstlane(wave);
    const bool ract = role < 3 && lane < half;
    const int d0 = min(lane, half - 1), d1 = d0 + half;
    ObTileStatsRt tr;
    _Float16 ur0 = (_Float16)0, ur1 = (_Float16)0, rc0 = (_Float16)0, rs0 = (_Float16)0, rc1 = (_Float16)0, rs1 = (_Float16)0;
    _Float16 br0 = (_Float16)0, br1 = (_Float16)0;
    const float *st_r = role == 0 ? A.st_q : (role == 1 ? A.st_k : A.st_v);
    const int n_r = role == 0 ? NQ : NK;
    if (PST) {
        const _Float16 *ub = role == 0 ? A.u_q + head * D : (role == 1 ? A.u_k + kvh * D : A.u_v + kvh * D);
        if (role < 3) {                                                           // (wave-uniform)
            ob_tiles_load_rt(tr, st_r, n_r, lane);
            ur0 = ub[d0]; ur1 = ub[d1];
            if (BIAS) {
                const _Float16 *bb = role == 0 ? A.b_q + head * D : (role == 1 ? A.b_k + kvh * D : A.b_v + kvh * D);
                br0 = bb[d0]; br1 = bb[d1];
            }
            if (A.rope_cur) { rc0 = A.rope_cur[d0]; rs0 = A.rope_cur[D + d0]; rc1 = A.rope_cur[d1]; rs1 = A.rope_cur[D + d1]; }
        }
    }
    const _Float16 cqh = PST ? (_Float16)0 : (ZIN ? ob_zg(A.z_q, A.g_q, 0) : A.u_q[0]), ckh = PST ? (_Float16)0 : (ZIN ? ob_zg(A.z_k, A.g_k, 0) : A.u_k[0]),
                   cvh = PST ? (_Float16)0 : (ZIN ? ob_zg(A.z_v, A.g_v, 0) : A.u_v[0]);
    const int dq = min(tid, D - 1), dp = dq < half ? dq + half : dq - half;     // own and rotate_half partner
    _Float16 uqh = (_Float16)0, ukh = (_Float16)0, uvh = (_Float16)0, uqp = (_Float16)0, ukp = (_Float16)0;
    _Float16 bqh = (_Float16)0, bkh = (_Float16)0, bvh = (_Float16)0, bqp = (_Float16)0, bkp = (_Float16)0;
    if (!PST && ZIN) {
        uqh = ob_zg(A.z_q, A.g_q, head * D + dq); ukh = ob_zg(A.z_k, A.g_k, kvh * D + dq); uvh = ob_zg(A.z_v, A.g_v, kvh * D + dq);
        uqp = ob_zg(A.z_q, A.g_q, head * D + dp); ukp = ob_zg(A.z_k, A.g_k, kvh * D + dp);
    }
    if (!PST) {
        if (!ZIN) {
            uqh = A.u_q[head * D + dq]; ukh = A.u_k[kvh * D + dq]; uvh = A.u_v[kvh * D + dq];
            uqp = A.u_q[head * D + dp]; ukp = A.u_k[kvh * D + dp];
        }
        if (BIAS) {
            bqh = A.b_q[head * D + dq]; bkh = A.b_k[kvh * D + dq]; bvh = A.b_v[kvh * D + dq];
            bqp = A.b_q[head * D + dp]; bkp = A.b_k[kvh * D + dp];
        }
    }
    const _Float16 hnx = A.h_next ? A.h_next[head * D + dq] : (_Float16)1;
    _Float16 cosh_ = (_Float16)0, sinh_ = (_Float16)0;
    if (!PST && A.rope_cur) { cosh_ = A.rope_cur[dq]; sinh_ = A.rope_cur[D + dq]; }      // requested with everything else (uniform branch)
    int pos_early = 0;
    if (!BLIND) {
        pos_early = *A.pos;
        if (pos_early < 0 || pos_early >= A.max_len) return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int64_t off = (int64_t)min(pg + PG * i, A.max_len - 1) * D + dcl;
        if (BLIND || pg + PG * i < pos_early) {   // rows >= pos are masked below either way (row pos comes from k_s / v_s)
            kreg[i] = *reinterpret_cast<const ob_half8 *>(kbase + off);
            vreg[i] = *reinterpret_cast<const ob_half8 *>(vbase + off);
        } else {
            kreg[i] = (ob_half8)(_Float16)0; vreg[i] = (ob_half8)(_Float16)0;
        }
    }
    // BLIND: no early return on the position -- hipcc sinks every load that is only used behind such a branch BELOW it
    // (the "blind" K / V requests were issued after the position had arrived: the dependent round trip they exist to
    // avoid).  An invalid position (a step past the cache: host error) computes on position 0 and stores nothing.
    const int pos_raw = BLIND ? *A.pos : pos_early;
    const bool live = pos_raw >= 0 && pos_raw < A.max_len;
    if (!BLIND && !live) return;                  // idle slot (batched step): uniform, before any barrier
    const int pos = live ? pos_raw : 0;
    const int L = pos + 1;
    if (!A.rope_cur) {
        if (PST) {
            if (role < 2) {
                rc0 = A.cos[(int64_t)pos * D + d0]; rs0 = A.sin[(int64_t)pos * D + d0];
                rc1 = A.cos[(int64_t)pos * D + d1]; rs1 = A.sin[(int64_t)pos * D + d1];
            }
        } else { cosh_ = A.cos[(int64_t)pos * D + dq]; sinh_ = A.sin[(int64_t)pos * D + dq]; }
    }
    __builtin_amdgcn_sched_barrier(0);

    // LayerNorm statistics of the three rows: from the producer's tile partials (every wave, no
    // barrier), or recomputed from the rows by each workgroup
    float mq = 0.f, rq = 0.f, mk = 0.f, rk = 0.f, mv = 0.f, rv = 0.f;
    if (PST) {
        if (role < 3) {                                                           // (wave-uniform)
            float mr, rr;
            ob_tiles_combine_rt(tr, st_r, n_r, A.ln_eps, lane, mr, rr);
            // apply_rotary_pos_emb (:175-181): x*cos + rotate_half(x)*sin, each op rounded to fp16; v: LayerNorm only
            float y0 = ob_ln_apply((float)ur0, mr, rr), y1 = ob_ln_apply((float)ur1, mr, rr);
            if (BIAS) { y0 = ob_round_h(y0 + (float)br0); y1 = ob_round_h(y1 + (float)br1); }     // output += bias (bitnet.py:119-120)
            float e0 = y0, e1 = y1;
            if (role < 2) {
                e0 = ob_round_h(ob_round_h(y0 * (float)rc0) + ob_round_h(-y1 * (float)rs0));
                e1 = ob_round_h(ob_round_h(y1 * (float)rc1) + ob_round_h(y0 * (float)rs1));
            }
            _Float16 *dst = role == 0 ? q_s : (role == 1 ? k_s : v_s);
            if (ract) {
                dst[d0] = (_Float16)e0; dst[d1] = (_Float16)e1;
                if (role > 0 && live && head % (H / Hkv) == 0) {      // one workgroup per kv head appends to the cache
                    _Float16 *cr = (role == 1 ? A.kcache : A.vcache) + ((int64_t)kvh * A.max_len + pos) * D;
                    cr[d0] = (_Float16)e0; cr[d1] = (_Float16)e1;
                }
            }
            for (int d = D + lane; d < 128; d += 64) dst[d] = (_Float16)0;     // zero padding beyond the head dimension
        }
    } else {
        const float cq = (float)cqh, ck = (float)ckh, cv = (float)cvh;
        ob_float2 a2[6] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        for (int base = tid * 8; base < NQ; base += NTH * 8)
            ob_stats8(ZIN ? ob_zg8(A.z_q, A.g_q, base) : *reinterpret_cast<const ob_half8 *>(A.u_q + base), cq, a2[0], a2[1]);
        for (int base = tid * 8; base < NK; base += NTH * 8) {
            ob_stats8(ZIN ? ob_zg8(A.z_k, A.g_k, base) : *reinterpret_cast<const ob_half8 *>(A.u_k + base), ck, a2[2], a2[3]);
            ob_stats8(ZIN ? ob_zg8(A.z_v, A.g_v, base) : *reinterpret_cast<const ob_half8 *>(A.u_v + base), cv, a2[4], a2[5]);
        }
        float s[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = a2[i][0] + a2[i][1];
        ob_block_sum_n<6, NWV>(s, red);                        // barrier 1
        ob_ln_stats(s[0], s[1], cq, NQ, A.ln_eps, mq, rq);
        ob_ln_stats(s[2], s[3], ck, NK, A.ln_eps, mk, rk);
        ob_ln_stats(s[4], s[5], cv, NK, A.ln_eps, mv, rv);
    }
    if (!PST && tid < 128) {
        float qe = 0.f, ke = 0.f, ve = 0.f;
        if (tid < D) {
            // apply_rotary_pos_emb (:175-181): q*cos + rotate_half(q)*sin, each op rounded to fp16
            const float c = (float)cosh_, sn = (float)sinh_;
            float q0 = ob_ln_apply((float)uqh, mq, rq), q1 = ob_ln_apply((float)uqp, mq, rq);
            float k0 = ob_ln_apply((float)ukh, mk, rk), k1 = ob_ln_apply((float)ukp, mk, rk);
            ve = ob_ln_apply((float)uvh, mv, rv);
            if (BIAS) {
                q0 = ob_round_h(q0 + (float)bqh); q1 = ob_round_h(q1 + (float)bqp);
                k0 = ob_round_h(k0 + (float)bkh); k1 = ob_round_h(k1 + (float)bkp);
                ve = ob_round_h(ve + (float)bvh);
            }
            const float qr = tid < half ? -q1 : q1, kr = tid < half ? -k1 : k1;
            qe = ob_round_h(ob_round_h(q0 * c) + ob_round_h(qr * sn));
            ke = ob_round_h(ob_round_h(k0 * c) + ob_round_h(kr * sn));
            if (live && head % (H / Hkv) == 0) {      // one workgroup per kv head appends to the cache
                A.kcache[((int64_t)kvh * A.max_len + pos) * D + tid] = (_Float16)ke;
                A.vcache[((int64_t)kvh * A.max_len + pos) * D + tid] = (_Float16)ve;
            }
        }
        q_s[tid] = (_Float16)qe; k_s[tid] = (_Float16)ke; v_s[tid] = (_Float16)ve;
    }
    __syncthreads();                                                 // barrier 2

    // scores: fp32-accumulated dot of fp16 pairs (v_dot2_f32_f16) -> fp16 (matmul output) -> / sqrt(D)
    // -> fp16 (:546).  Positions < 128 stay in registers, later ones go through `sc`.
    const float inv_sqrt_d = __builtin_amdgcn_rsqf((float)D);      // scalar divisor: multiply by the reciprocal
    auto dot8 = [](const ob_half8 a, const ob_half8 b) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const ob_half2 x = {a[2 * e], a[2 * e + 1]}, y = {b[2 * e], b[2 * e + 1]};
            acc = __builtin_amdgcn_fdot2(x, y, acc, false);
        }
        return acc;
    };
    const ob_half8 q8 = *reinterpret_cast<const ob_half8 *>(q_s + 8 * ds);       // zero beyond D
    const ob_half8 kn8 = *reinterpret_cast<const ob_half8 *>(k_s + 8 * ds);
    const ob_half8 vn8 = *reinterpret_cast<const ob_half8 *>(v_s + 8 * ds);
    float sreg[NI];
    float lmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = pg + PG * i;
        sreg[i] = -INFINITY;
        // (batched step: whole sweeps beyond the sequence are skipped -- uniform per workgroup; with heads x slots
        // workgroups sharing the SIMDs every masked instruction is somebody else's issue slot.  The single-sequence kernel
        // computes all eight sweeps: skipping them by uniform branches measured SLOWER, 5.0 -> 5.5 us per launch, round 4)
        if (BLIND || PG * i < L) {
            const float dot = ob_row_sum(dot8(q8, p == pos ? kn8 : kreg[i]));
            const float sv = ob_round_h(ob_round_h(dot) * inv_sqrt_d);
            sreg[i] = p < L ? sv : -INFINITY;
            lmax = fmaxf(lmax, sreg[i]);
        }
    }
    for (int p0 = NB; p0 < L; p0 += NB) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = p0 + pg + PG * i;
            if (p < L) {
                const ob_half8 k8 = p == pos ? kn8 : *reinterpret_cast<const ob_half8 *>(kbase + (int64_t)p * D + dcl);
                const float dot = ob_row_sum(dot8(q8, k8));
                const float sv = ob_round_h(ob_round_h(dot) * inv_sqrt_d);
                if (ds == 0) sc[p] = sv;
                lmax = fmaxf(lmax, sv);
            }
        }
    }
    // softmax in fp32 (:562), probabilities rounded to fp16
    lmax = ob_rows_max(lmax);
    if (lane == 0) red[96 + wave] = lmax;
    __syncthreads();                                                 // barrier 3
    float gmax;
    {
        const ob_float4 m0 = *reinterpret_cast<const ob_float4 *>(red + 96);
        gmax = fmaxf(fmaxf(m0[0], m0[1]), fmaxf(m0[2], m0[3]));
        if (NWV == 8) {
            const ob_float4 m1 = *reinterpret_cast<const ob_float4 *>(red + 100);
            gmax = fmaxf(gmax, fmaxf(fmaxf(m1[0], m1[1]), fmaxf(m1[2], m1[3])));
        }
    }
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (BLIND || PG * i < L) { sreg[i] = __expf(sreg[i] - gmax); lsum += sreg[i]; }      // exp(-inf) = 0
        else sreg[i] = 0.f;
    }
    for (int p0 = NB; p0 < L; p0 += NB) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = p0 + pg + PG * i;
            if (p < L) {                       // sc[p] was written by lane ds == 0 of this same row
                const float e = __expf(sc[p] - gmax);
                if (ds == 0) sc[p] = e;
                lsum += e;
            }
        }
    }
    lsum = ob_rows_sum(lsum);
    if (lane == 0) red[112 + wave] = lsum;
    __syncthreads();                                                 // barrier 4
    float inv_l;
    {
        const ob_float4 l0 = *reinterpret_cast<const ob_float4 *>(red + 112);
        float lt = (l0[0] + l0[1]) + (l0[2] + l0[3]);
        if (NWV == 8) {
            const ob_float4 l1 = *reinterpret_cast<const ob_float4 *>(red + 116);
            lt += (l1[0] + l1[1]) + (l1[2] + l1[3]);
        }
        inv_l = 1.0f / lt;
    }
    // out = P . V over this thread's positions and 8 dims
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = pg + PG * i;
        if (BLIND || PG * i < L) {
            const float pr = ob_round_h(sreg[i] * inv_l);
            const ob_half8 vv = p == pos ? vn8 : vreg[i];
            if (p < L) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += pr * (float)vv[e];
            }
        }
    }
    for (int p0 = NB; p0 < L; p0 += NB) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = p0 + pg + PG * i;
            if (p < L) {
                const float pr = ob_round_h(sc[p] * inv_l);
                const ob_half8 vv = p == pos ? vn8 : *reinterpret_cast<const ob_half8 *>(vbase + (int64_t)p * D + dcl);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += pr * (float)vv[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ob_rows_sum(o[e]);
    if (lane < 16) {
        float *dst = po + wave * 128 + 8 * ds;
        *reinterpret_cast<ob_float4 *>(dst) = (ob_float4){o[0], o[1], o[2], o[3]};
        *reinterpret_cast<ob_float4 *>(dst + 4) = (ob_float4){o[4], o[5], o[6], o[7]};
    }
    __syncthreads();                                                 // barrier 5
    if (tid < D) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) acc += po[w * 128 + tid];
        _Float16 oh = (_Float16)acc;
        if (A.h_next) oh = oh * hnx;
        if (live) A.out[head * D + tid] = oh;
    }
}

// ---------------------------------------------------------------------------------------------
// Long contexts: the same attention split over the positions (one workgroup per head reads the whole
// KV history at ~15 GB/s; at 2k tokens that is 70 us per layer).  Two launches over (head, split):
//   scores kernel: LayerNorm(q, k) + RoPE of the new token (recomputed per workgroup, as above), KV
//                  append by (leader head, split 0), scores of the split's positions -> scratch, and
//                  the split's (max, sum of exp relative to that max);
//   pv kernel:     global max and denominator from the split statistics (fp32; the denominator is
//                  sum_s l_s * exp(m_s - max)), p = fp16(exp(score - max) / l) exactly as the single
//                  kernel forms it, partial P.V of the split (fp32) -> scratch; the LAST workgroup of
//                  a head to arrive (agent-scope counter) adds the partials in split order and
//                  writes the fp16 output -- deterministic.
// Scratch per head: scores [max_len] fp32 | stats [S][2] | partial o [S][128] | counter.
// ---------------------------------------------------------------------------------------------
struct ObAttnSplitArgs {
    ObAttnArgs a;
    float *scores;        // [H][max_len]
    float *stats;         // [H][S][2]
    float *part;          // [H][S][128]
    int *counter;         // [H], zero before the first use; the combining workgroup resets it
    int S, chunk;         // splits, positions per split
};

__global__ __launch_bounds__(OB_ATTN_THREADS) void ob_dec_attn_scores_kernel(const ObAttnSplitArgs B)
{
    const ObAttnArgs &A = B.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = A.D, H = A.H, Hkv = A.Hkv;
    const int head = blockIdx.x, split = blockIdx.y, kvh = head / (H / Hkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *red = reinterpret_cast<float *>(smem);
    _Float16 *q_s = reinterpret_cast<_Float16 *>(smem + 512);
    _Float16 *k_s = q_s + 128;
    float *sl = reinterpret_cast<float *>(k_s + 128);        // [chunk] this split's scores (second pass)
    const int pos = *A.pos;
    if (pos < 0 || pos >= A.max_len) return;                 // a step past the cache: touch nothing
    const int L = pos + 1;
    const int p_lo = split * B.chunk, p_hi = min(L, p_lo + B.chunk);
    if (p_lo >= L) return;                                   // empty split (uniform)
    const int NQ = H * D, NK = Hkv * D;
    const _Float16 *kbase = A.kcache + (int64_t)kvh * A.max_len * D;
    const int ds = tid & 15, pg = tid >> 4;
    const int dcl = 8 * ds < D ? 8 * ds : 0;
    const _Float16 cqh = A.u_q[0], ckh = A.u_k[0], cvh = A.u_v[0];
    const int half = D >> 1;
    const int dq = min(tid, D - 1), dp = dq < half ? dq + half : dq - half;
    const _Float16 uqh = A.u_q[head * D + dq], ukh = A.u_k[kvh * D + dq], uvh = A.u_v[kvh * D + dq];
    const _Float16 uqp = A.u_q[head * D + dp], ukp = A.u_k[kvh * D + dp];
    const _Float16 cosh_ = A.cos[(int64_t)pos * D + dq], sinh_ = A.sin[(int64_t)pos * D + dq];
    const bool has_b = A.b_q != nullptr;                     // config.attention_bias (uniform)
    const _Float16 bqh = has_b ? A.b_q[head * D + dq] : (_Float16)0, bkh = has_b ? A.b_k[kvh * D + dq] : (_Float16)0,
                   bvh = has_b ? A.b_v[kvh * D + dq] : (_Float16)0, bqp = has_b ? A.b_q[head * D + dp] : (_Float16)0,
                   bkp = has_b ? A.b_k[kvh * D + dp] : (_Float16)0;
    float mq, rq, mk, rk, mv, rv;
    if (A.st_q) {                                            // producer's tile partials (uniform branch)
        ObTileStatsRt tq, tk, tv;
        ob_tiles_load_rt(tq, A.st_q, NQ, lane); ob_tiles_load_rt(tk, A.st_k, NK, lane); ob_tiles_load_rt(tv, A.st_v, NK, lane);
        ob_tiles_combine_rt(tq, A.st_q, NQ, A.ln_eps, lane, mq, rq);
        ob_tiles_combine_rt(tk, A.st_k, NK, A.ln_eps, lane, mk, rk);
        ob_tiles_combine_rt(tv, A.st_v, NK, A.ln_eps, lane, mv, rv);
    } else {
        const float cq = (float)cqh, ck = (float)ckh, cv = (float)cvh;
        ob_float2 a2[6] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        for (int base = tid * 8; base < NQ; base += OB_ATTN_THREADS * 8)
            ob_stats8(*reinterpret_cast<const ob_half8 *>(A.u_q + base), cq, a2[0], a2[1]);
        for (int base = tid * 8; base < NK; base += OB_ATTN_THREADS * 8) {
            ob_stats8(*reinterpret_cast<const ob_half8 *>(A.u_k + base), ck, a2[2], a2[3]);
            ob_stats8(*reinterpret_cast<const ob_half8 *>(A.u_v + base), cv, a2[4], a2[5]);
        }
        float s[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = a2[i][0] + a2[i][1];
        ob_block_sum_n<6, OB_ATTN_WAVES>(s, red);
        ob_ln_stats(s[0], s[1], cq, NQ, A.ln_eps, mq, rq);
        ob_ln_stats(s[2], s[3], ck, NK, A.ln_eps, mk, rk);
        ob_ln_stats(s[4], s[5], cv, NK, A.ln_eps, mv, rv);
    }
    if (tid < 128) {
        float qe = 0.f, ke = 0.f;
        if (tid < D) {
            const float c = (float)cosh_, sn = (float)sinh_;
            float q0 = ob_ln_apply((float)uqh, mq, rq), q1 = ob_ln_apply((float)uqp, mq, rq);
            float k0 = ob_ln_apply((float)ukh, mk, rk), k1 = ob_ln_apply((float)ukp, mk, rk);
            float ve = ob_ln_apply((float)uvh, mv, rv);
            if (has_b) {                                     // output += bias (bitnet.py:119-120), one fp16 rounding each
                q0 = ob_round_h(q0 + (float)bqh); q1 = ob_round_h(q1 + (float)bqp);
                k0 = ob_round_h(k0 + (float)bkh); k1 = ob_round_h(k1 + (float)bkp);
                ve = ob_round_h(ve + (float)bvh);
            }
            const float qr = tid < half ? -q1 : q1, kr = tid < half ? -k1 : k1;
            qe = ob_round_h(ob_round_h(q0 * c) + ob_round_h(qr * sn));
            ke = ob_round_h(ob_round_h(k0 * c) + ob_round_h(kr * sn));
            if (split == 0 && head % (H / Hkv) == 0) {       // one workgroup per kv head appends to the cache
                A.kcache[((int64_t)kvh * A.max_len + pos) * D + tid] = (_Float16)ke;
                A.vcache[((int64_t)kvh * A.max_len + pos) * D + tid] = (_Float16)ve;
            }
        }
        q_s[tid] = (_Float16)qe; k_s[tid] = (_Float16)ke;
    }
    __syncthreads();
    const float inv_sqrt_d = __builtin_amdgcn_rsqf((float)D);
    auto dot8 = [](const ob_half8 a, const ob_half8 b) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const ob_half2 x = {a[2 * e], a[2 * e + 1]}, y = {b[2 * e], b[2 * e + 1]};
            acc = __builtin_amdgcn_fdot2(x, y, acc, false);
        }
        return acc;
    };
    const ob_half8 q8 = *reinterpret_cast<const ob_half8 *>(q_s + 8 * ds);
    const ob_half8 kn8 = *reinterpret_cast<const ob_half8 *>(k_s + 8 * ds);
    float *sc = B.scores + (int64_t)head * A.max_len;
    float lmax = -INFINITY;
    for (int p0 = p_lo; p0 < p_hi; p0 += 128) {
        ob_half8 k8[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)                            // the block's four rows in flight together
            k8[i] = *reinterpret_cast<const ob_half8 *>(kbase + (int64_t)min(p0 + pg + 32 * i, A.max_len - 1) * D + dcl);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + pg + 32 * i;
            const float dot = ob_row_sum(dot8(q8, p == pos ? kn8 : k8[i]));
            const float sv = ob_round_h(ob_round_h(dot) * inv_sqrt_d);
            if (p < p_hi) {
                if (ds == 0) { sc[p] = sv; sl[p - p_lo] = sv; }
                lmax = fmaxf(lmax, sv);
            }
        }
    }
    lmax = ob_rows_max(lmax);
    if (lane == 0) red[96 + wave] = lmax;
    __syncthreads();
    float m;
    {
        const ob_float4 m0 = *reinterpret_cast<const ob_float4 *>(red + 96), m1 = *reinterpret_cast<const ob_float4 *>(red + 100);
        m = fmaxf(fmaxf(fmaxf(m0[0], m0[1]), fmaxf(m0[2], m0[3])), fmaxf(fmaxf(m1[0], m1[1]), fmaxf(m1[2], m1[3])));
    }
    float lsum = 0.f;                                          // LDS copy, written before the barrier above
    for (int p0 = p_lo; p0 < p_hi; p0 += 128) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + pg + 32 * i;
            if (p < p_hi) lsum += __expf(sl[p - p_lo] - m);
        }
    }
    lsum = ob_rows_sum(lsum);
    if (lane == 0) red[112 + wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        const ob_float4 l0 = *reinterpret_cast<const ob_float4 *>(red + 112), l1 = *reinterpret_cast<const ob_float4 *>(red + 116);
        float *st = B.stats + ((int64_t)head * B.S + split) * 2;
        st[0] = m;
        st[1] = ((l0[0] + l0[1]) + (l0[2] + l0[3])) + ((l1[0] + l1[1]) + (l1[2] + l1[3]));
    }
}

__global__ __launch_bounds__(OB_ATTN_THREADS) void ob_dec_attn_pv_kernel(const ObAttnSplitArgs B)
{
    const ObAttnArgs &A = B.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = A.D, H = A.H, Hkv = A.Hkv;
    const int head = blockIdx.x, split = blockIdx.y, kvh = head / (H / Hkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *po = reinterpret_cast<float *>(smem);             // [8 waves][128]
    int *flag = reinterpret_cast<int *>(po + OB_ATTN_WAVES * 128);
    const int pos = *A.pos;                                  // the position has NOT been advanced yet
    if (pos < 0 || pos >= A.max_len) return;
    const int L = pos + 1;
    const int p_lo = split * B.chunk, p_hi = min(L, p_lo + B.chunk);
    if (p_lo >= L) return;
    const int nsplit = (L + B.chunk - 1) / B.chunk;          // splits that hold positions
    const _Float16 *vbase = A.vcache + (int64_t)kvh * A.max_len * D;
    const int ds = tid & 15, pg = tid >> 4;
    const int dcl = 8 * ds < D ? 8 * ds : 0;
    // global max and denominator from the split statistics (every thread, a few L2 reads)
    const float *st = B.stats + (int64_t)head * B.S * 2;
    ob_float2 ml[16];                                        // all splits' (max, sum) in one round trip (S <= 16)
#pragma unroll
    for (int j = 0; j < 16; ++j) ml[j] = *reinterpret_cast<const ob_float2 *>(st + 2 * min(j, nsplit - 1));
    float gmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) gmax = fmaxf(gmax, ml[j][0]);          // clamped duplicates do not change the maximum
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) l += j < nsplit ? ml[j][1] * __expf(ml[j][0] - gmax) : 0.f;
    const float inv_l = 1.0f / l;
    const float *sc = B.scores + (int64_t)head * A.max_len;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p0 = p_lo; p0 < p_hi; p0 += 128) {
        ob_half8 v8[4];
        float pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = min(p0 + pg + 32 * i, A.max_len - 1);
            v8[i] = *reinterpret_cast<const ob_half8 *>(vbase + (int64_t)p * D + dcl);
            pr[i] = sc[min(p, L - 1)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + pg + 32 * i;
            if (p < p_hi) {
                const float w = ob_round_h(__expf(pr[i] - gmax) * inv_l);     // softmax in fp32, probability -> fp16 (:562)
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += w * (float)v8[i][e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ob_rows_sum(o[e]);
    if (lane < 16) {
        float *dst = po + wave * 128 + 8 * ds;
        *reinterpret_cast<ob_float4 *>(dst) = (ob_float4){o[0], o[1], o[2], o[3]};
        *reinterpret_cast<ob_float4 *>(dst + 4) = (ob_float4){o[4], o[5], o[6], o[7]};
    }
    __syncthreads();
    float *part = B.part + ((int64_t)head * B.S) * 128;
    if (tid < 128) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < OB_ATTN_WAVES; ++w) acc += po[w * 128 + tid];
        part[split * 128 + tid] = acc;
    }
    __syncthreads();                                          // the workgroup's stores are ordered before ...
    if (tid == 0) {                                           // ... ONE agent-scope release / acquire (L2 write-back, invalidate)
        const int prev = __hip_atomic_fetch_add(B.counter + head, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        *flag = prev == nsplit - 1;
        if (prev == nsplit - 1) __hip_atomic_store(B.counter + head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!*flag) return;
    if (tid < D) {                                            // partials are read past the caches (agent-scope loads)
        float v[16];                                           // all partials in flight together, then a fixed-order sum
#pragma unroll
        for (int j = 0; j < 16; ++j)
            v[j] = __hip_atomic_load(part + min(j, nsplit - 1) * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += j < nsplit ? v[j] : 0.f;
        A.out[head * D + tid] = (_Float16)acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Final norm + fp16 lm_head GEMV + per-workgroup argmax (modeling_bitllama.py:1321,1610-1611;
// generation/utils.py:2540).  Persistent grid, one row per wave iteration.
// ---------------------------------------------------------------------------------------------
struct ObHeadArgs {
    const _Float16 *hres_in, *u_prev, *rms_w;   // residual stream, last down_proj pre-LN, final norm weight
    const float *st_prev;                       // optional: tile partials of u_prev from the down_proj GEMV
    const _Float16 *lm_w;                       // [V, K]
    _Float16 *logits;                           // [V] fp16 (the reference's logits before .float())
    float *part_val; int *part_idx;             // [grid] per-workgroup argmax
    _Float16 *hres_out;                         // final hidden state (post residual), optional
    int K, V;
    float rms_eps, ln_eps;
};

__global__ __launch_bounds__(OB_DEC_THREADS) void ob_dec_lmhead_kernel(const ObHeadArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *x_s = reinterpret_cast<_Float16 *>(smem);                  // [K]
    float *red = reinterpret_cast<float *>(smem + (size_t)A.K * 2);      // 64 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = A.K;
    // prologue: h = hres + LN(u_prev); x = RMSNorm(h) * w
    {
        float hv[OB_DEC_MAXV][8], uv[OB_DEC_MAXV][8];
        const float c = (float)A.u_prev[0];
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
                const ob_half8 tu = *reinterpret_cast<const ob_half8 *>(A.u_prev + base);
                const ob_half8 th = *reinterpret_cast<const ob_half8 *>(A.hres_in + base);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uv[v][i] = (float)tu[i]; hv[v][i] = (float)th[i];
                    const float d = uv[v][i] - c; s[0] += d; s[1] += d * d;
                }
            }
        }
        float mean, rstd;
        if (A.st_prev) {
            ObTileStatsRt tp;
            ob_tiles_load_rt(tp, A.st_prev, K, lane);
            ob_tiles_combine_rt(tp, A.st_prev, K, A.ln_eps, lane, mean, rstd);
        } else {
            ob_block_sum_n<2, OB_DEC_WAVES>(s, red);
            ob_ln_stats(s[0], s[1], c, K, A.ln_eps, mean, rstd);
        }
        float ss[1] = {0.f};
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    hv[v][i] = ob_round_h(hv[v][i] + ob_ln_apply(uv[v][i], mean, rstd));
                    ss[0] += hv[v][i] * hv[v][i];
                }
            }
        }
        ob_block_sum_n<1, OB_DEC_WAVES>(ss, red + 32);
        const float rs = __builtin_amdgcn_rsqf(ss[0] * __builtin_amdgcn_rcpf((float)K) + A.rms_eps);
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
                const ob_half8 wv = *reinterpret_cast<const ob_half8 *>(A.rms_w + base);
                ob_half8 xo, ho;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ho[i] = (_Float16)hv[v][i];
                    xo[i] = (_Float16)((float)wv[i] * ob_round_h(hv[v][i] * rs));
                }
                *reinterpret_cast<ob_half8 *>(x_s + base) = xo;
                if (blockIdx.x == 0 && A.hres_out) *reinterpret_cast<ob_half8 *>(A.hres_out + base) = ho;
            }
        }
    }
    __syncthreads();
    // rows: two rows per wave iteration, 4 x 16-byte loads per row in flight per lane
    float best = -INFINITY;
    int besti = 0x7fffffff;
    const int gw = blockIdx.x * OB_DEC_WAVES + wave, nw = gridDim.x * OB_DEC_WAVES;
    for (int row = gw; row < A.V; row += 2 * nw) {
        const int rowb = row + nw;
        const bool hasb = rowb < A.V;
        const _Float16 *wa = A.lm_w + (int64_t)row * K;
        const _Float16 *wb = A.lm_w + (int64_t)(hasb ? rowb : row) * K;
        float acca = 0.f, accb = 0.f;
        for (int k0 = lane * 8; k0 < K; k0 += 2048) {
            ob_half8 ra[4], rb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = min(k0 + 512 * i, K - 8);
                ra[i] = __builtin_nontemporal_load(reinterpret_cast<const ob_half8 *>(wa + k));
                rb[i] = __builtin_nontemporal_load(reinterpret_cast<const ob_half8 *>(wb + k));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 512 * i;
                if (k < K) {
                    const ob_half8 xv = *reinterpret_cast<const ob_half8 *>(x_s + k);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { acca += (float)ra[i][e] * (float)xv[e]; accb += (float)rb[i][e] * (float)xv[e]; }
                }
            }
        }
        acca = ob_wave_sum(acca);
        accb = ob_wave_sum(accb);
        const float la = ob_round_h(acca), lb = ob_round_h(accb);
        if (lane == 0) {
            A.logits[row] = (_Float16)la;
            if (hasb) A.logits[rowb] = (_Float16)lb;
        }
        if (la > best || (la == best && row < besti)) { best = la; besti = row; }
        if (hasb && (lb > best || (lb == best && rowb < besti))) { best = lb; besti = rowb; }
    }
    // workgroup argmax (first index on ties)
    __syncthreads();
    if (lane == 0) { red[wave] = best; reinterpret_cast<int *>(red)[16 + wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        float bv = red[0]; int bi = reinterpret_cast<int *>(red)[16];
        for (int w = 1; w < OB_DEC_WAVES; ++w) {
            const float v = red[w]; const int i = reinterpret_cast<int *>(red)[16 + w];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
        A.part_val[blockIdx.x] = bv;
        A.part_idx[blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(256) void ob_dec_argmax_kernel(const float *part_val, const int *part_idx, int nparts,
                                                            int *token, int *pos, int *out_tokens, int max_out, int vocab)
{
    __shared__ float sv[256];
    __shared__ int si[256];
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        const float v = part_val[i]; const int ix = part_idx[i];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float v = sv[threadIdx.x + off]; const int ix = si[threadIdx.x + off];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ix < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = ix; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int p = *pos;
        // all-NaN logits leave the index at its initial value: clamp, or the next replay would read
        // embed[token] out of bounds (no host in the loop under graph replay)
        const int tk = (si[0] >= 0 && si[0] < vocab) ? si[0] : 0;
        *token = tk;
        if (out_tokens && p < max_out) out_tokens[p] = tk;
        *pos = p + 1;
    }
}
