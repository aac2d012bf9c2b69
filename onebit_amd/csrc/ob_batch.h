// Batched decode step (one new token for each of B sequences): the row-wise glue kernels around the
// skinny 1-bit GEMM (ob_skinny.h).  Same arithmetic and rounding points as the prologues of
// ob_dec_gemv_kernel (which does them per workgroup for ONE sequence): here they run once per row,
// and the packed weights of a projection are streamed once per step for all sequences.
//   ob_b_norm_kernel<EMBED>  r = embed[token] | hres_in + LayerNorm(u_prev);  x = RMSNorm(r) * w
//                            (modeling_bitllama.py:912-918, 76-81)
//   ob_b_swiglu_kernel       act = silu(LayerNorm(u_gate)) * LayerNorm(u_up)   (:257)
// Attention reuses ob_dec_attn_kernel with blockIdx.y = sequence slot.
#pragma once
#include "ob_decode.h"

struct ObBNormArgs {
    const _Float16 *embed;        // EMBED: [vocab, H]
    const int *tokens;            // EMBED: [B]
    const _Float16 *hres_in;      // !EMBED: [B, H]
    const _Float16 *u_prev;       // !EMBED: [B, H] pre-LayerNorm output of the previous projection, or NULL with
    const float *z0, *z1;         //   fp32 split-K partial sums [B, H] of it (u = fp16(fp16(z0 + z1) * g_prev))
    const _Float16 *g_prev;       //   and its weight_scale [H]
    const _Float16 *rms_w;        // [H]
    _Float16 *hres_out;           // [B, H]
    _Float16 *x;                  // [B, H] (may be NULL when only the scaled outputs are wanted)
    int H;
    float rms_eps, ln_eps;
    // up to 3 consumers' pre-scaled activations a_i = fp16(x * h_i) (the rounding of bitnet.py:113 done by the
    // producer: the consuming projections are then called with ONEBIT_FLAG_PRESCALED and skip their own pass)
    const _Float16 *h_next[3];
    _Float16 *x_scaled[3];
    int n_scaled;
    ObPfPlan pf;                  // optional (nseg > 0): packed rows of the next GEMM launch to pull into L2 (ob_common.h) --
    int pf_rows;                  //   by the workgroups beyond the first pf_rows (= rows) of the grid, which do nothing else
    // (appended in round 5: the fields above keep their kernarg offsets)
    const _Float16 *bias_prev;    // optional [H]: bias of the projection that produced u_prev (o_proj with config.attention_bias):
                                  //   r = hres_in + fp16(LayerNorm(u_prev) + bias_prev)  (bitnet.py:119-120, then :912)
    // (appended in round 6) optional [grid]: workgroup t READS row rows[t] of hres_in / u_prev and writes row t of the outputs --
    const int *rows;              //   the final norm of a mixed step on the rows whose logits are wanted (onebit_mixed_step)
    const float *z2, *z3;         // optional third / fourth partial sum beside z0, z1 (K-slices: ob_gemm3_ksplit)
};

// NV = 8-half vectors per thread actually populated: ceil(H / 4096).  (Sized OB_DEC_MAXV = 4 for every width, a 4096-wide
// row executed four times the loads, reductions and stores it needed, three quarters of them masked.)
template <bool EMBED, int NV>
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_b_norm_kernel(const ObBNormArgs A)
{
    __shared__ __attribute__((aligned(16))) float red[128];
    const int tid = threadIdx.x, H = A.H;
    if (ob_prefetch_only_wg(A.pf, A.pf_rows, tid, OB_DEC_THREADS)) return;
    const int64_t row = (int64_t)blockIdx.x * H;                       // row written
    const int64_t rin = (!EMBED && A.rows) ? (int64_t)A.rows[blockIdx.x] * H : row;       // row read (uniform pointer test)
    const _Float16 *src = EMBED ? A.embed + (int64_t)A.tokens[blockIdx.x] * H : A.hres_in + rin;
    ob_half8 hv[NV], uv[NV];
    // the vectors of the LAST phase (RMSNorm weight, the consumers' input_factor) are requested here, with the rows: asked
    // for after the two block reductions, their L2 round trip was the tail of every launch
    ob_half8 wv[NV], hn[3][NV];
    bool valid[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        valid[v] = base < H;
        hv[v] = *reinterpret_cast<const ob_half8 *>(src + (valid[v] ? base : 0));
        wv[v] = *reinterpret_cast<const ob_half8 *>(A.rms_w + (valid[v] ? base : 0));
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < A.n_scaled) hn[j][v] = *reinterpret_cast<const ob_half8 *>(A.h_next[j] + (valid[v] ? base : 0));
        if (!EMBED) {
            const int b0 = valid[v] ? base : 0;
            if (A.u_prev) {
                uv[v] = *reinterpret_cast<const ob_half8 *>(A.u_prev + rin + b0);
            } else {                                // split-K partials: the epilogue of bitnet.py:115-116 happens here
                const ob_half8 gv = *reinterpret_cast<const ob_half8 *>(A.g_prev + b0);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const ob_float4 a = *reinterpret_cast<const ob_float4 *>(A.z0 + rin + b0 + 4 * q);
                    // (z1 == NULL: one complete sum, the all-reduced partials of a K-sharded projection -- uniform test)
                    const ob_float4 b = A.z1 ? *reinterpret_cast<const ob_float4 *>(A.z1 + rin + b0 + 4 * q) : (ob_float4){0.f, 0.f, 0.f, 0.f};
                    ob_float4 ab = a + b;
                    if (A.z2) ab = ab + *reinterpret_cast<const ob_float4 *>(A.z2 + rin + b0 + 4 * q);
                    if (A.z3) ab = ab + *reinterpret_cast<const ob_float4 *>(A.z3 + rin + b0 + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        uv[v][4 * q + i] = (_Float16)(ob_round_h(ab[i]) * (float)gv[4 * q + i]);
                }
            }
        }
    }
    if (!EMBED) {
        // pivot of the shifted sums = element 0 of the row, the same value in every thread
        const float c0 = A.u_prev ? (float)A.u_prev[rin]
                                  : (float)(_Float16)(ob_round_h(((A.z0[rin] + (A.z1 ? A.z1[rin] : 0.f)) + (A.z2 ? A.z2[rin] : 0.f)) + (A.z3 ? A.z3[rin] : 0.f)) * (float)A.g_prev[0]);
        ob_float2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (valid[v]) ob_stats8(uv[v], c0, s2, q2);
        float s[2] = {s2[0] + s2[1], q2[0] + q2[1]};
        ob_block_sum_n<2, OB_DEC_WAVES>(s, red);
        float mean, rstd;
        ob_ln_stats(s[0], s[1], c0, H, A.ln_eps, mean, rstd);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            ob_half8 ln;
#pragma unroll
            for (int i = 0; i < 8; ++i) ln[i] = ob_ln_apply_h(uv[v][i], rstd, nmr);
            if (A.bias_prev && valid[v])         // (uniform pointer test; batched decode step of a checkpoint with attention_bias)
                ln = ln + *reinterpret_cast<const ob_half8 *>(A.bias_prev + (v * OB_DEC_THREADS + tid) * 8);
            hv[v] = hv[v] + ln;                  // residual + hidden_states
        }
    }
    float ss[1] = {0.f};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (valid[v]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const ob_half2 pr = {hv[v][2 * i], hv[v][2 * i + 1]};
                ss[0] = __builtin_amdgcn_fdot2(pr, pr, ss[0], false);
            }
        }
    }
    ob_block_sum_n<1, OB_DEC_WAVES>(ss, red + 64);
    const float rs = __builtin_amdgcn_rsqf(ss[0] * __builtin_amdgcn_rcpf((float)H) + A.rms_eps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        if (valid[v]) {
            ob_half8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (_Float16)__builtin_fmaf((float)hv[v][i], rs, 0.0f);
            const ob_half8 xv = wv[v] * t;
            if (A.x) *reinterpret_cast<ob_half8 *>(A.x + row + base) = xv;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < A.n_scaled) *reinterpret_cast<ob_half8 *>(A.x_scaled[j] + row + base) = xv * hn[j][v];
            *reinterpret_cast<ob_half8 *>(A.hres_out + row + base) = hv[v];
        }
    }
}

// The LayerNorm that ends a BitLinearInf (bitnet.py:118-120) as its own pass over fp16 rows u [T, N] -- onebit_linear_forward without
// ONEBIT_FLAG_SKIP_LN, i.e. the module path of a prefill call: y = fp16(LayerNorm(u)) (+ bias, rounded again).  Round 6: the row kernels'
// form (packed halves in registers, ONE shifted-sum reduction, v_fma_mixlo) instead of ob_layernorm_rows_kernel's fp32 copy of the row
// and two reductions: the module path of a [16384, 4096] -> 11008 call 1.235 -> 1.208 ms (tools/module_prefill_probe.py).  In place (y == u) is fine: every thread reads its elements and the pivot before the
// reduction's barrier and writes after it.
// ZIN: the rows arrive as up to four fp32 K-slice sums z0 .. z3 [T, N] of the LDS-DMA GEMM's K-sliced form (ob_gemm3_ksplit) and the weight
// scale g: u = fp16(fp16(z0 + z1 + ..) * g) (bitnet.py:115-116) is formed here; `skip` (ONEBIT_FLAG_SKIP_LN) then writes u itself.
struct ObLnRowsArgs {
    const _Float16 *uin; const float *z[4]; const _Float16 *g, *bias; _Float16 *y;
    int N, skip; float eps;
};
template <int NV, bool BIAS, bool ZIN = false>
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_ln_rows_kernel(const ObLnRowsArgs A)
{
    __shared__ __attribute__((aligned(16))) float red[32];
    const int tid = threadIdx.x, N = A.N;
    const int64_t r0 = (int64_t)blockIdx.x * N;
    _Float16 *out = A.y + r0;
    ob_half8 u[NV], bv[BIAS ? NV : 1];
    bool ok[NV];
    auto zg8 = [&](int base) -> ob_half8 {
        ob_float4 a = *reinterpret_cast<const ob_float4 *>(A.z[0] + r0 + base), b = *reinterpret_cast<const ob_float4 *>(A.z[0] + r0 + base + 4);
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (A.z[j]) { a = a + *reinterpret_cast<const ob_float4 *>(A.z[j] + r0 + base); b = b + *reinterpret_cast<const ob_float4 *>(A.z[j] + r0 + base + 4); }
        const ob_half8 gv = *reinterpret_cast<const ob_half8 *>(A.g + base);
        ob_half8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = (_Float16)(ob_round_h(a[i]) * (float)gv[i]);
            o[4 + i] = (_Float16)(ob_round_h(b[i]) * (float)gv[4 + i]);
        }
        return o;
    };
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        ok[v] = base < N;
        if (ZIN) u[v] = zg8(ok[v] ? base : 0);
        else u[v] = *reinterpret_cast<const ob_half8 *>(A.uin + r0 + (ok[v] ? base : 0));
        if (BIAS) bv[v] = *reinterpret_cast<const ob_half8 *>(A.bias + (ok[v] ? base : 0));
    }
    if (ZIN && A.skip) {                                     // (uniform) the pre-LayerNorm rows themselves
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (ok[v]) *reinterpret_cast<ob_half8 *>(out + (v * OB_DEC_THREADS + tid) * 8) = u[v];
        return;
    }
    float c0;
    if (ZIN) {
        float z = A.z[0][r0];
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (A.z[j]) z += A.z[j][r0];
        c0 = (float)(_Float16)(ob_round_h(z) * (float)A.g[0]);
    } else c0 = (float)A.uin[r0];
    ob_float2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (ok[v]) ob_stats8(u[v], c0, s2, q2);
    float st[2] = {s2[0] + s2[1], q2[0] + q2[1]};
    ob_block_sum_n<2, OB_DEC_WAVES>(st, red);
    float mean, rstd;
    ob_ln_stats(st[0], st[1], c0, N, A.eps, mean, rstd);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (!ok[v]) continue;
        ob_half8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = ob_ln_apply_h(u[v][i], rstd, nmr);
        if (BIAS) o = o + bv[v];
        *reinterpret_cast<ob_half8 *>(out + (v * OB_DEC_THREADS + tid) * 8) = o;
    }
}

// u = fp16(fp16(z) * g) (bitnet.py:115-116) for up to three vectors of COMPLETE fp32 sums z (the all-reduced partials of
// K-sharded projections: onebit_decode_step_ksharded), optionally with the per-16-row-tile LayerNorm partials (sum, M2) the
// decode kernels' PST forms combine.  Block b covers 4096 elements of its segment; a tile is a pair of adjacent lanes.
struct ObBZgSeg { const float *z; const _Float16 *g; _Float16 *u; float *st; int n, blk_end; };
struct ObBZgArgs { ObBZgSeg s[3]; };
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_b_zg_kernel(const ObBZgArgs A)
{
    const int b = (int)blockIdx.x;
    const int si = (b >= A.s[0].blk_end ? 1 : 0) + (b >= A.s[1].blk_end ? 1 : 0);
    const ObBZgSeg S = si == 0 ? A.s[0] : (si == 1 ? A.s[1] : A.s[2]);
    const int b0 = si == 0 ? 0 : (si == 1 ? A.s[0].blk_end : A.s[1].blk_end);
    const int base = ((b - b0) * OB_DEC_THREADS + (int)threadIdx.x) * 8;
    if (base >= S.n) return;                                 // n % 8 == 0 (n % 16 == 0 with st): lane pairs leave together
    const ob_float4 z0 = *reinterpret_cast<const ob_float4 *>(S.z + base), z1 = *reinterpret_cast<const ob_float4 *>(S.z + base + 4);
    const ob_half8 g = *reinterpret_cast<const ob_half8 *>(S.g + base);
    ob_half8 u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[i] = (_Float16)(ob_round_h(z0[i]) * (float)g[i]);
        u[4 + i] = (_Float16)(ob_round_h(z1[i]) * (float)g[4 + i]);
    }
    *reinterpret_cast<ob_half8 *>(S.u + base) = u;
    if (S.st) {
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sm += (float)u[i];
        sm += OB_DPP_F(sm, 0xB1, 0xF);                       // lane ^ 1: the other half of the 16-row tile
        const float mean = sm * 0.0625f;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = (float)u[i] - mean; m2 += d * d; }
        m2 += OB_DPP_F(m2, 0xB1, 0xF);
        if ((threadIdx.x & 1) == 0) *reinterpret_cast<ob_float2 *>(S.st + 2 * (base >> 4)) = (ob_float2){sm, m2};
    }
}

struct ObBSwigluArgs {
    const _Float16 *u_gate, *u_up;   // [B, I]
    _Float16 *act;                   // [B, I]
    int I;
    float ln_eps;
    const _Float16 *h_next;          // optional: act <- fp16(act * h_next), the consumer's pre-scaled activations
    const float *ext;                // optional [B, 4] {mean, rstd} of the COMPLETE gate and up rows: the rows given here are a
                                     // column slice (tensor-parallel N-shard), their statistics were combined across ranks
    ObPfPlan pf;                     // optional (nseg > 0): packed rows of the next GEMM launch to pull into L2, by the
    int pf_rows;                     //   workgroups beyond the first pf_rows (= rows) of the grid
    // (appended in round 5) K-sharded decode step: the rows as COMPLETE fp32 sums + weight_scale instead of fp16 u --
    // u = fp16(fp16(z) * g) (bitnet.py:115-116) is formed here; u_gate / u_up are then not read
    const float *z_gate, *z_up;      // [B, I]
    const _Float16 *g_gate, *g_up;   // [I]
};

template <int NV>
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_b_swiglu_kernel(const ObBSwigluArgs A)
{
    __shared__ __attribute__((aligned(16))) float red[128];
    const int tid = threadIdx.x, I = A.I;
    if (ob_prefetch_only_wg(A.pf, A.pf_rows, tid, OB_DEC_THREADS)) return;
    const int64_t row = (int64_t)blockIdx.x * I;
    ob_half8 g8[NV], u8[NV], hn[NV];
    bool valid[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        valid[v] = base < I;
        if (A.z_gate) {                                          // (uniform) reduced sums of a K-sharded projection
            const int b0 = valid[v] ? base : 0;
            const ob_half8 gg = *reinterpret_cast<const ob_half8 *>(A.g_gate + b0), gu = *reinterpret_cast<const ob_half8 *>(A.g_up + b0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const ob_float4 zg = *reinterpret_cast<const ob_float4 *>(A.z_gate + row + b0 + 4 * q);
                const ob_float4 zu = *reinterpret_cast<const ob_float4 *>(A.z_up + row + b0 + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    g8[v][4 * q + i] = (_Float16)(ob_round_h(zg[i]) * (float)gg[4 * q + i]);
                    u8[v][4 * q + i] = (_Float16)(ob_round_h(zu[i]) * (float)gu[4 * q + i]);
                }
            }
        } else {
            g8[v] = *reinterpret_cast<const ob_half8 *>(A.u_gate + row + (valid[v] ? base : 0));
            u8[v] = *reinterpret_cast<const ob_half8 *>(A.u_up + row + (valid[v] ? base : 0));
        }
        if (A.h_next) hn[v] = *reinterpret_cast<const ob_half8 *>(A.h_next + (valid[v] ? base : 0));   // (used after the reduction)
    }
    float mg, rg, mu, ru;
    if (A.ext) {                                            // uniform per launch
        const ob_float4 e = *reinterpret_cast<const ob_float4 *>(A.ext + (size_t)blockIdx.x * 4);
        mg = e[0]; rg = e[1]; mu = e[2]; ru = e[3];
    } else {
        const float c0 = A.z_gate ? (float)(_Float16)(ob_round_h(A.z_gate[row]) * (float)A.g_gate[0]) : (float)A.u_gate[row];
        const float c1 = A.z_gate ? (float)(_Float16)(ob_round_h(A.z_up[row]) * (float)A.g_up[0]) : (float)A.u_up[row];
        ob_float2 sg2 = {0.f, 0.f}, qg2 = {0.f, 0.f}, su2 = {0.f, 0.f}, qu2 = {0.f, 0.f};
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (valid[v]) { ob_stats8(g8[v], c0, sg2, qg2); ob_stats8(u8[v], c1, su2, qu2); }
        float s[4] = {sg2[0] + sg2[1], qg2[0] + qg2[1], su2[0] + su2[1], qu2[0] + qu2[1]};
        ob_block_sum_n<4, OB_DEC_WAVES>(s, red);
        ob_ln_stats(s[0], s[1], c0, I, A.ln_eps, mg, rg);
        ob_ln_stats(s[2], s[3], c1, I, A.ln_eps, mu, ru);
    }
    const float ng = -mg * rg, nu = -mu * ru;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        if (valid[v]) {
            ob_half8 sg, up;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const _Float16 gh = ob_ln_apply_h(g8[v][i], rg, ng);
                up[i] = ob_ln_apply_h(u8[v][i], ru, nu);
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf((float)gh, -1.44269504088896341f, 0.0f));
                sg[i] = (_Float16)__builtin_fmaf((float)gh, __builtin_amdgcn_rcpf(1.0f + e), 0.0f);
            }
            ob_half8 av = sg * up;                                               // act_fn(gate) * up -> fp16
            if (A.h_next) av = av * hn[v];                                                      // fp16(act * h), bitnet.py:113
            *reinterpret_cast<ob_half8 *>(A.act + row + base) = av;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Prefill glue between the q|k|v projections and the attention kernel, one workgroup per token row:
// LayerNorm of the three pre-LayerNorm rows (bitnet.py:118 -- the projections are called with
// ONEBIT_FLAG_SKIP_LN), RoPE on q and k (apply_rotary_pos_emb, modeling_bitllama.py:175-181: q*cos +
// rotate_half(q)*sin, every op rounded to fp16) and the [B, S, heads, D] -> [B, heads, S, D] transpose
// (:526-528) in one pass: q goes to its own [B, H, S, D] tensor, k and v straight into the KV cache rows
// [slot][kv head][past + s][D].  Replaces one LayerNorm kernel per projection plus ~8 elementwise torch
// kernels (cat / mul / add / neg / copy: 37 ms of a 262 ms 7B prefill of 8 x 2048 tokens).
// ---------------------------------------------------------------------------------------------
struct ObQkvRopeArgs {
    const _Float16 *u_q, *u_k, *u_v;     // [T, H*D], [T, Hkv*D], [T, Hkv*D] pre-LayerNorm, T = B * S
    const _Float16 *cos, *sin;           // [max_pos, D]
    _Float16 *q;                         // [B, H, S, D], or [B, S, H, D] with q_bshd (the caller then hands the
                                         // attention kernel a transposed view and gets its output in token-major rows)
    _Float16 *kcache, *vcache;           // [slots, Hkv, max_len, D]
    int S, H, Hkv, D, past, max_len, q_bshd;
    float ln_eps;
    const float *ext;                    // optional [T, 6] {mean, rstd} of the COMPLETE q, k, v rows (tensor-parallel: the rows
                                         // given here hold the rank's heads only)
    // (round 6) RAGGED rows -- the token rows of several sequences in one launch (mixed prefill + decode step, long-context
    // decode steps): row t belongs to cache slot row_slot[t] (NULL: slot t) and sits at position row_pos[t]; q stays
    // token-major.  A row whose slot / position lies outside [0, n_slots) x [0, max_len) is skipped (idle decode slot).
    const int *row_slot, *row_pos;       // device [T]; row_pos != NULL selects this form (S, past unused)
    int n_slots;
    // (round 6) config.attention_bias (modeling_bitllama.py:451-453): q / k / v = fp16(LayerNorm(u) + b) (bitnet.py:118-120)
    // BEFORE the rotary embedding; all three or none
    const _Float16 *b_q, *b_k, *b_v;     // [H*D], [Hkv*D], [Hkv*D]
};

// BIAS: a compile-time switch -- with the bias path as a run-time branch the bias-free kernel carried five more register arrays and ran
// 70 % longer at 4120 rows x 5120 (82 vs 47 us: profiles/r06_mixed_step_kernels.txt, first version)
template <int NV, bool BIAS = false>
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_qkv_rope_kernel(const ObQkvRopeArgs A)
{
    __shared__ __attribute__((aligned(16))) float red[128];
    const int tid = threadIdx.x, D = A.D, half = D >> 1;
    const int NQ = A.H * D, NK = A.Hkv * D;
    const int t = blockIdx.x;
    int b, sp, pos;
    if (A.row_pos) {                                        // ragged rows (uniform pointer test)
        b = A.row_slot ? A.row_slot[t] : t; pos = A.row_pos[t]; sp = 0;
        if (b < 0 || b >= A.n_slots || pos < 0 || pos >= A.max_len) return;      // idle row: the whole workgroup leaves
    } else {
        b = t / A.S; sp = t - b * A.S; pos = A.past + sp;
    }
    const _Float16 *uq = A.u_q + (int64_t)t * NQ, *uk = A.u_k + (int64_t)t * NK, *uv = A.u_v + (int64_t)t * NK;
    // thread owns 8 consecutive elements of each vector per pass (D % 8 == 0: never straddles a head), and
    // fetches their rotate_half partners (same head, d +- D/2) alongside
    ob_half8 q8[NV], qp8[NV], k8[NV], kp8[NV], v8[NV];
    bool vq[NV], vk[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        vq[v] = base < NQ; vk[v] = base < NK;
        const int bq = vq[v] ? base : 0, bk = vk[v] ? base : 0;
        const int dq = bq % D, dk = bk % D;
        q8[v] = *reinterpret_cast<const ob_half8 *>(uq + bq);
        k8[v] = *reinterpret_cast<const ob_half8 *>(uk + bk);
        v8[v] = *reinterpret_cast<const ob_half8 *>(uv + bk);
        (void)dq; (void)dk;
    }
    // rotate_half partners: elements d +- D/2 of the same head live D/16 lanes away (8 elements per lane, heads are
    // aligned groups of D/8 lanes) -- a lane exchange instead of a second pass over the rows.  Lanes beyond the
    // vector hold clamped copies of lane 0's chunk and exchange among themselves.
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const ob_u32x4 qa = __builtin_bit_cast(ob_u32x4, q8[v]), ka = __builtin_bit_cast(ob_u32x4, k8[v]);
        ob_u32x4 qb, kb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qb[i] = (uint32_t)__shfl_xor((int)qa[i], D >> 4);
            kb[i] = (uint32_t)__shfl_xor((int)ka[i], D >> 4);
        }
        qp8[v] = __builtin_bit_cast(ob_half8, qb);
        kp8[v] = __builtin_bit_cast(ob_half8, kb);
    }
    constexpr bool has_b = BIAS;
    ob_half8 bq8[BIAS ? NV : 1], bqp8[BIAS ? NV : 1], bk8[BIAS ? NV : 1], bkp8[BIAS ? NV : 1], bv8[BIAS ? NV : 1];
    if constexpr (BIAS) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            const int bq = vq[v] ? base : 0, bk = vk[v] ? base : 0;
            bq8[v] = *reinterpret_cast<const ob_half8 *>(A.b_q + bq);
            bk8[v] = *reinterpret_cast<const ob_half8 *>(A.b_k + bk);
            bv8[v] = *reinterpret_cast<const ob_half8 *>(A.b_v + bk);
            const ob_u32x4 qa = __builtin_bit_cast(ob_u32x4, bq8[v]), ka = __builtin_bit_cast(ob_u32x4, bk8[v]);
            ob_u32x4 qb, kb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qb[i] = (uint32_t)__shfl_xor((int)qa[i], D >> 4);
                kb[i] = (uint32_t)__shfl_xor((int)ka[i], D >> 4);
            }
            bqp8[v] = __builtin_bit_cast(ob_half8, qb);
            bkp8[v] = __builtin_bit_cast(ob_half8, kb);
        }
    }
    float mq, rq, mk, rk, mv, rv;
    if (A.ext) {                                            // uniform per launch
        const float *e = A.ext + (size_t)t * 6;
        mq = e[0]; rq = e[1]; mk = e[2]; rk = e[3]; mv = e[4]; rv = e[5];
    } else {
        const float cq = (float)uq[0], ck = (float)uk[0], cv = (float)uv[0];
        ob_float2 a2[6] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (vq[v]) ob_stats8(q8[v], cq, a2[0], a2[1]);
            if (vk[v]) { ob_stats8(k8[v], ck, a2[2], a2[3]); ob_stats8(v8[v], cv, a2[4], a2[5]); }
        }
        float st[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) st[i] = a2[i][0] + a2[i][1];
        ob_block_sum_n<6, OB_DEC_WAVES>(st, red);
        ob_ln_stats(st[0], st[1], cq, NQ, A.ln_eps, mq, rq);
        ob_ln_stats(st[2], st[3], ck, NK, A.ln_eps, mk, rk);
        ob_ln_stats(st[4], st[5], cv, NK, A.ln_eps, mv, rv);
    }
    const _Float16 *cosr = A.cos + (int64_t)pos * D, *sinr = A.sin + (int64_t)pos * D;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        if (vq[v]) {
            const int hd = base / D, d0 = base - hd * D;
            const ob_half8 c8 = *reinterpret_cast<const ob_half8 *>(cosr + d0), s8 = *reinterpret_cast<const ob_half8 *>(sinr + d0);
            ob_half8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x0 = ob_ln_apply((float)q8[v][i], mq, rq), x1 = ob_ln_apply((float)qp8[v][i], mq, rq);
                if constexpr (BIAS) { x0 = ob_round_h(x0 + (float)bq8[v][i]); x1 = ob_round_h(x1 + (float)bqp8[v][i]); }
                const float xr = d0 < half ? -x1 : x1;
                o[i] = (_Float16)ob_round_h(ob_round_h(x0 * (float)c8[i]) + ob_round_h(xr * (float)s8[i]));
            }
            _Float16 *qd = (A.q_bshd || A.row_pos) ? A.q + (int64_t)t * NQ + base : A.q + (((int64_t)b * A.H + hd) * A.S + sp) * D + d0;
            *reinterpret_cast<ob_half8 *>(qd) = o;
        }
        if (vk[v]) {
            const int hd = base / D, d0 = base - hd * D;
            const ob_half8 c8 = *reinterpret_cast<const ob_half8 *>(cosr + d0), s8 = *reinterpret_cast<const ob_half8 *>(sinr + d0);
            ob_half8 ok, ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x0 = ob_ln_apply((float)k8[v][i], mk, rk), x1 = ob_ln_apply((float)kp8[v][i], mk, rk);
                float xv = ob_ln_apply((float)v8[v][i], mv, rv);
                if constexpr (BIAS) {
                    x0 = ob_round_h(x0 + (float)bk8[v][i]); x1 = ob_round_h(x1 + (float)bkp8[v][i]);
                    xv = ob_round_h(xv + (float)bv8[v][i]);
                }
                const float xr = d0 < half ? -x1 : x1;
                ok[i] = (_Float16)ob_round_h(ob_round_h(x0 * (float)c8[i]) + ob_round_h(xr * (float)s8[i]));
                ov[i] = (_Float16)xv;
            }
            const int64_t off = (((int64_t)b * A.Hkv + hd) * A.max_len + pos) * D + d0;
            *reinterpret_cast<ob_half8 *>(A.kcache + off) = ok;
            *reinterpret_cast<ob_half8 *>(A.vcache + off) = ov;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Batched lm_head + greedy sampling (modeling_bitllama.py:1610-1611, generation/utils.py:2540):
// logits[b][v] = fp16( sum_k x[b][k] * W[v][k] ), next[b] = argmax_v (first index on ties).
// The fp16 lm_head matrix (2 V H bytes: 262 MB at 7B) is streamed ONCE per step for all B <= 64
// sequences: a workgroup owns 128 vocabulary rows (8 waves x 16 rows); K advances in chunks of 256
// whose activations [B, 256] all threads stage into LDS (32 KB, double buffered, one barrier per
// chunk) while each lane streams its row's 512 bytes of the chunk straight into MFMA A operands
// (v_mfma_f32_16x16x32_f16, weights = A, tokens = B operand, as everywhere).  HBM-bound: the MFMA work
// is 2 instructions per KB of weights.  Per-workgroup (max, index) per sequence go to scratch; a
// second tiny kernel reduces them over the workgroups.
// ---------------------------------------------------------------------------------------------
struct ObBHeadArgs {
    const _Float16 *x;            // [B, H] final-norm output
    const _Float16 *lm_w;         // [V, H]
    _Float16 *logits;             // [B, V] or NULL
    float *part_val;              // [grid][64]
    int *part_idx;                // [grid][64]
    int B, H, V;
};
#define OB_BH_ROWS 128
#define OB_BH_K 256
#define OB_BH_PITCH (OB_BH_K + 8)            // halves per LDS row (528 B)

template <int TT>                            // token tiles of 16: B <= 16 * TT
__global__ __launch_bounds__(512) void ob_b_lmhead_kernel(const ObBHeadArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 (*xs)[TT * 16][OB_BH_PITCH] = reinterpret_cast<_Float16 (*)[TT * 16][OB_BH_PITCH]>(smem);
    float *rv = reinterpret_cast<float *>(smem + (size_t)2 * TT * 16 * OB_BH_PITCH * 2);      // [8 waves][TT*16]
    int *ri = reinterpret_cast<int *>(rv + 8 * TT * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, gq = lane >> 4;
    const int H = A.H, B = A.B, V = A.V;
    const int v0 = blockIdx.x * OB_BH_ROWS + wave * 16;
    const _Float16 *wrow = A.lm_w + (int64_t)min(v0 + r, V - 1) * H + gq * 16;       // lane: 16 consecutive k per 64-k block
    const int nkc = (H + OB_BH_K - 1) / OB_BH_K;

    // staging: thread -> (token st_b + 16 i, halves st_k .. st_k + 7) of a [TT*16, 256] chunk
    const int st_b = tid >> 5, st_k = (tid & 31) * 8;
    auto stage = [&](int kc, int buf) {
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const int b = st_b + 16 * i, k = kc * OB_BH_K + st_k;
            ob_half8 v = (ob_half8)(_Float16)0;
            if (b < B && k < H) v = *reinterpret_cast<const ob_half8 *>(A.x + (int64_t)b * H + k);
            *reinterpret_cast<ob_half8 *>(&xs[buf][b][st_k]) = v;
        }
    };
    ob_float4 acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = (ob_float4){0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    __syncthreads();
    for (int kc = 0; kc < nkc; ++kc) {
        const int buf = kc & 1;
        // this lane's 8 x 16 bytes of its row for the chunk (non-temporal: read exactly once per step)
        ob_half8 wf[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = kc * OB_BH_K + q * 64;
            const int kk = min(k, H - 64);                       // H % 64 == 0 (host-checked); clamp the tail chunk
            wf[2 * q] = __builtin_nontemporal_load(reinterpret_cast<const ob_half8 *>(wrow + kk));
            wf[2 * q + 1] = __builtin_nontemporal_load(reinterpret_cast<const ob_half8 *>(wrow + kk + 8));
        }
        if (kc + 1 < nkc) stage(kc + 1, buf ^ 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (kc * OB_BH_K + q * 64 < H) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        const ob_half8 bx = *reinterpret_cast<const ob_half8 *>(&xs[buf][t * 16 + r][q * 64 + gq * 16 + 8 * s]);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[2 * q + s], bx, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    // D[v][b]: lane holds v = v0 + 4 gq + i (i = 0..3), b = t * 16 + r
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int b = t * 16 + r;
        float best = -INFINITY;
        int besti = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = v0 + 4 * gq + i;
            const float lg = ob_round_h(acc[t][i]);                // logits are fp16 tensors in the reference
            if (v < V && (lg > best || (lg == best && v < besti))) { best = lg; besti = v; }
        }
        if (A.logits && b < B) {
            const int vb = v0 + 4 * gq;
            if (vb + 3 < V && (V & 3) == 0) {
                const ob_half4 o = {(_Float16)acc[t][0], (_Float16)acc[t][1], (_Float16)acc[t][2], (_Float16)acc[t][3]};
                *reinterpret_cast<ob_half4 *>(A.logits + (int64_t)b * V + vb) = o;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (vb + i < V) A.logits[(int64_t)b * V + vb + i] = (_Float16)acc[t][i];
            }
        }
        // over the 4 lane groups (rows of 16 lanes) of the wave: the two gfx950 row swaps
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            uint32_t ov, oi;
            if (step == 0) {
                auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                auto c = __builtin_amdgcn_permlane16_swap((uint32_t)besti, (uint32_t)besti, false, false);
                // with both operands equal the swap returns {own value, partner row's value} in some order:
                // the XOR of the pair with our own value is the partner's
                ov = a[0] ^ a[1] ^ __float_as_uint(best); oi = c[0] ^ c[1] ^ (uint32_t)besti;
            } else {
                auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                auto c = __builtin_amdgcn_permlane32_swap((uint32_t)besti, (uint32_t)besti, false, false);
                ov = a[0] ^ a[1] ^ __float_as_uint(best); oi = c[0] ^ c[1] ^ (uint32_t)besti;
            }
            const float pv = __uint_as_float(ov);
            const int pi = (int)oi;
            if (pv > best || (pv == best && pi < besti)) { best = pv; besti = pi; }
        }
        if (gq == 0) { rv[wave * TT * 16 + b] = best; ri[wave * TT * 16 + b] = besti; }
    }
    __syncthreads();
    if (tid < TT * 16) {
        float bv = rv[tid];
        int bi = ri[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float v = rv[w * TT * 16 + tid];
            const int i = ri[w * TT * 16 + tid];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
        A.part_val[blockIdx.x * 64 + tid] = bv;
        A.part_idx[blockIdx.x * 64 + tid] = bi;
    }
}

// next[b] = argmax over the workgroup partials (first index on ties); one workgroup per sequence
__global__ __launch_bounds__(256) void ob_b_argmax_kernel(const float *part_val, const int *part_idx, int nparts, int vocab,
                                                          int *next_tokens)
{
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        const float v = part_val[i * 64 + b];
        const int ix = part_idx[i * 64 + b];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float v = sv[threadIdx.x + off];
            const int ix = si[threadIdx.x + off];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ix < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = ix; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) next_tokens[b] = (si[0] >= 0 && si[0] < vocab) ? si[0] : 0;      // all-NaN logits: clamp
}
