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
    _Float16 *x;                  // [B, H]
    int H;
    float rms_eps, ln_eps;
};

template <bool EMBED>
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_b_norm_kernel(const ObBNormArgs A)
{
    __shared__ __attribute__((aligned(16))) float red[128];
    const int tid = threadIdx.x, H = A.H;
    const int64_t row = (int64_t)blockIdx.x * H;
    const _Float16 *src = EMBED ? A.embed + (int64_t)A.tokens[blockIdx.x] * H : A.hres_in + row;
    ob_half8 hv[OB_DEC_MAXV], uv[OB_DEC_MAXV];
    bool valid[OB_DEC_MAXV];
#pragma unroll
    for (int v = 0; v < OB_DEC_MAXV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        valid[v] = base < H;
        hv[v] = *reinterpret_cast<const ob_half8 *>(src + (valid[v] ? base : 0));
        if (!EMBED) {
            const int b0 = valid[v] ? base : 0;
            if (A.u_prev) {
                uv[v] = *reinterpret_cast<const ob_half8 *>(A.u_prev + row + b0);
            } else {                                // split-K partials: the epilogue of bitnet.py:115-116 happens here
                const ob_half8 gv = *reinterpret_cast<const ob_half8 *>(A.g_prev + b0);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const ob_float4 a = *reinterpret_cast<const ob_float4 *>(A.z0 + row + b0 + 4 * q);
                    const ob_float4 b = *reinterpret_cast<const ob_float4 *>(A.z1 + row + b0 + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        uv[v][4 * q + i] = (_Float16)(ob_round_h(a[i] + b[i]) * (float)gv[4 * q + i]);
                }
            }
        }
    }
    if (!EMBED) {
        // pivot of the shifted sums = element 0 of the row, the same value in every thread
        const float c0 = A.u_prev ? (float)A.u_prev[row]
                                  : (float)(_Float16)(ob_round_h(A.z0[row] + A.z1[row]) * (float)A.g_prev[0]);
        ob_float2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v)
            if (valid[v]) ob_stats8(uv[v], c0, s2, q2);
        float s[2] = {s2[0] + s2[1], q2[0] + q2[1]};
        ob_block_sum_n<2, OB_DEC_WAVES>(s, red);
        float mean, rstd;
        ob_ln_stats(s[0], s[1], c0, H, A.ln_eps, mean, rstd);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            ob_half8 ln;
#pragma unroll
            for (int i = 0; i < 8; ++i) ln[i] = ob_ln_apply_h(uv[v][i], rstd, nmr);
            hv[v] = hv[v] + ln;                  // residual + hidden_states
        }
    }
    float ss[1] = {0.f};
#pragma unroll
    for (int v = 0; v < OB_DEC_MAXV; ++v) {
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
    for (int v = 0; v < OB_DEC_MAXV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        if (valid[v]) {
            const ob_half8 w = *reinterpret_cast<const ob_half8 *>(A.rms_w + base);
            ob_half8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (_Float16)__builtin_fmaf((float)hv[v][i], rs, 0.0f);
            *reinterpret_cast<ob_half8 *>(A.x + row + base) = w * t;
            *reinterpret_cast<ob_half8 *>(A.hres_out + row + base) = hv[v];
        }
    }
}

struct ObBSwigluArgs {
    const _Float16 *u_gate, *u_up;   // [B, I]
    _Float16 *act;                   // [B, I]
    int I;
    float ln_eps;
};

__global__ __launch_bounds__(OB_DEC_THREADS) void ob_b_swiglu_kernel(const ObBSwigluArgs A)
{
    __shared__ __attribute__((aligned(16))) float red[128];
    const int tid = threadIdx.x, I = A.I;
    const int64_t row = (int64_t)blockIdx.x * I;
    ob_half8 g8[OB_DEC_MAXV], u8[OB_DEC_MAXV];
    bool valid[OB_DEC_MAXV];
#pragma unroll
    for (int v = 0; v < OB_DEC_MAXV; ++v) {
        const int base = (v * OB_DEC_THREADS + tid) * 8;
        valid[v] = base < I;
        g8[v] = *reinterpret_cast<const ob_half8 *>(A.u_gate + row + (valid[v] ? base : 0));
        u8[v] = *reinterpret_cast<const ob_half8 *>(A.u_up + row + (valid[v] ? base : 0));
    }
    const float c0 = (float)A.u_gate[row], c1 = (float)A.u_up[row];
    ob_float2 sg2 = {0.f, 0.f}, qg2 = {0.f, 0.f}, su2 = {0.f, 0.f}, qu2 = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < OB_DEC_MAXV; ++v)
        if (valid[v]) { ob_stats8(g8[v], c0, sg2, qg2); ob_stats8(u8[v], c1, su2, qu2); }
    float s[4] = {sg2[0] + sg2[1], qg2[0] + qg2[1], su2[0] + su2[1], qu2[0] + qu2[1]};
    ob_block_sum_n<4, OB_DEC_WAVES>(s, red);
    float mg, rg, mu, ru;
    ob_ln_stats(s[0], s[1], c0, I, A.ln_eps, mg, rg);
    ob_ln_stats(s[2], s[3], c1, I, A.ln_eps, mu, ru);
    const float ng = -mg * rg, nu = -mu * ru;
#pragma unroll
    for (int v = 0; v < OB_DEC_MAXV; ++v) {
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
            *reinterpret_cast<ob_half8 *>(A.act + row + base) = sg * up;
        }
    }
}
