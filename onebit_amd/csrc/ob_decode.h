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
#include "ob_common.h"

#define OB_DEC_THREADS 512
#define OB_DEC_WAVES 8
#define OB_DEC_MAXV 4            // per-thread vectors of 8 halves: vector widths up to 16384

struct ObProj {
    const uint32_t *w;           // packed signs [N, ldw words]
    const _Float16 *h;           // input_factor [K]
    const _Float16 *g;           // weight_scale [N]
    _Float16 *u;                 // out: pre-LayerNorm u [N]
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
    float rms_eps, ln_eps;
};

template <int NV>
__device__ __forceinline__ void ob_block_sum_n(float (&v)[NV], float *red)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ob_wave_sum(v[i]);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[i * 16 + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[i * 16 + w];
        v[i] = s;
    }
}

// LayerNorm statistics (mean, rstd) from shifted sums s1 = sum(u - c), s2 = sum((u - c)^2).
__device__ __forceinline__ void ob_ln_stats(float s1, float s2, float c, int n, float eps, float &mean,
                                            float &rstd)
{
    const float m1 = s1 / (float)n;
    mean = c + m1;
    const float var = fmaxf(s2 / (float)n - m1 * m1, 0.f);
    rstd = 1.0f / sqrtf(var + eps);
}

__device__ __forceinline__ float ob_ln_apply(float u, float mean, float rstd)
{
    return ob_round_h((u - mean) * rstd);
}

__device__ __forceinline__ float ob_silu_h(float x)   // fp16 silu: fp32 math, one rounding
{
    return ob_round_h(x / (1.0f + __expf(-x)));
}

// ---------------------------------------------------------------------------------------------
// Prologue: builds x[K] (fp16 values held as float in xs[][]), then a_p = fp16(x * h_p) in LDS.
// lds_a layout: [nproj][Kpad] halves, Kpad = K rounded up to 512 (zero padded).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ob_dec_prologue(const ObGemvArgs &A, _Float16 *lds_a, float *red, int Kpad)
{
    const int tid = threadIdx.x;
    const int K = A.K;
    float xs[OB_DEC_MAXV][8];
    if (A.prologue == OB_P_PLAIN) {
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
                const ob_half8 t = *reinterpret_cast<const ob_half8 *>(A.xin + base);
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[v][i] = (float)t[i];
            }
        }
    } else if (A.prologue == OB_P_SWIGLU) {
        float gs[OB_DEC_MAXV][8], us[OB_DEC_MAXV][8];
        const float cg = (float)A.u_gate[0], cu = (float)A.u_up[0];
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
                const ob_half8 tg = *reinterpret_cast<const ob_half8 *>(A.u_gate + base);
                const ob_half8 tu = *reinterpret_cast<const ob_half8 *>(A.u_up + base);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    gs[v][i] = (float)tg[i];
                    us[v][i] = (float)tu[i];
                    const float dg = gs[v][i] - cg, du = us[v][i] - cu;
                    s[0] += dg; s[1] += dg * dg; s[2] += du; s[3] += du * du;
                }
            }
        }
        ob_block_sum_n<4>(s, red);
        float mg, rg, mu, ru;
        ob_ln_stats(s[0], s[1], cg, K, A.ln_eps, mg, rg);
        ob_ln_stats(s[2], s[3], cu, K, A.ln_eps, mu, ru);
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float gate = ob_ln_apply(gs[v][i], mg, rg);
                    const float up = ob_ln_apply(us[v][i], mu, ru);
                    xs[v][i] = ob_round_h(ob_silu_h(gate) * up);     // act_fn(gate) * up, modeling_bitllama.py:257
                }
            }
        }
    } else {
        // residual stream: embedding row, or h_in + LayerNorm(u_prev); then RMSNorm
        float hv[OB_DEC_MAXV][8];
        if (A.prologue == OB_P_EMBED_RMS) {
            const _Float16 *row = A.embed + (int64_t)(*A.token) * K;
#pragma unroll
            for (int v = 0; v < OB_DEC_MAXV; ++v) {
                const int base = (v * OB_DEC_THREADS + tid) * 8;
                if (base < K) {
                    const ob_half8 t = *reinterpret_cast<const ob_half8 *>(row + base);
#pragma unroll
                    for (int i = 0; i < 8; ++i) hv[v][i] = (float)t[i];
                }
            }
        } else {
            float uv[OB_DEC_MAXV][8];
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
                        uv[v][i] = (float)tu[i];
                        hv[v][i] = (float)th[i];
                        const float d = uv[v][i] - c;
                        s[0] += d; s[1] += d * d;
                    }
                }
            }
            ob_block_sum_n<2>(s, red);
            float mean, rstd;
            ob_ln_stats(s[0], s[1], c, K, A.ln_eps, mean, rstd);
#pragma unroll
            for (int v = 0; v < OB_DEC_MAXV; ++v) {
                const int base = (v * OB_DEC_THREADS + tid) * 8;
                if (base < K) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)       // residual + hidden_states, modeling_bitllama.py:912,918
                        hv[v][i] = ob_round_h(hv[v][i] + ob_ln_apply(uv[v][i], mean, rstd));
                }
            }
        }
        // RMSNorm (modeling_bitllama.py:76-81): fp32 variance, x * rsqrt -> fp16, * weight -> fp16
        float ss[1] = {0.f};
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
#pragma unroll
                for (int i = 0; i < 8; ++i) ss[0] += hv[v][i] * hv[v][i];
            }
        }
        ob_block_sum_n<1>(ss, red);
        const float rs = rsqrtf(ss[0] / (float)K + A.rms_eps);
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
                const ob_half8 wv = *reinterpret_cast<const ob_half8 *>(A.rms_w + base);
                ob_half8 ho;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ho[i] = (_Float16)hv[v][i];
                    xs[v][i] = ob_round_h((float)wv[i] * ob_round_h(hv[v][i] * rs));
                }
                if (blockIdx.x == 0 && A.hres_out)
                    *reinterpret_cast<ob_half8 *>(A.hres_out + base) = ho;
            }
        }
    }
    // a_p = fp16(x * h_p)  (bitnet.py:113), zero padding up to Kpad
    for (int p = 0; p < A.nproj; ++p) {
        _Float16 *dst = lds_a + (size_t)p * Kpad;
#pragma unroll
        for (int v = 0; v < OB_DEC_MAXV; ++v) {
            const int base = (v * OB_DEC_THREADS + tid) * 8;
            if (base < K) {
                const ob_half8 hv8 = *reinterpret_cast<const ob_half8 *>(A.p[p].h + base);
                ob_half8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (_Float16)(xs[v][i] * (float)hv8[i]);
                *reinterpret_cast<ob_half8 *>(dst + base) = o;
            } else if (base < Kpad) {
                *reinterpret_cast<ob_half8 *>(dst + base) = (ob_half8)(_Float16)0;
            }
        }
    }
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

__device__ __forceinline__ ob_u32x4 ob_dec_load_w(const ObProj &P, int row0, int chunk, int lane)
{
    const int r = lane & 15, gq = lane >> 4;
    const int row = min(row0 + r, P.N - 1);
    const int word = chunk * 16 + gq * 4;
    const int nwords = P.K >> 5;
    const uint32_t *src = P.w + (int64_t)row * P.ldw + word;
    ob_u32x4 w4 = {0u, 0u, 0u, 0u};
    if (word + 4 <= nwords && (P.ldw & 3) == 0) {
        w4 = __builtin_nontemporal_load(reinterpret_cast<const ob_u32x4 *>(src));
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (word + q < nwords) w4[q] = src[q];
    }
    return w4;
}

// dynamic LDS: [nproj * Kpad halves][MT * 8 waves * 16 rows floats][128 floats]
// PT = max 512-weight chunks per wave per tile (ceil(K/4096)), MT = max tiles per workgroup; both
// compile-time so that the in-flight weight registers and the accumulators are statically indexed.
template <int PT, int MT>
__global__ __launch_bounds__(OB_DEC_THREADS) void ob_dec_gemv_kernel(const ObGemvArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = A.K;
    const int Kpad = (K + 511) & ~511;
    _Float16 *lds_a = reinterpret_cast<_Float16 *>(smem);
    float *lds_red = reinterpret_cast<float *>(smem + (size_t)A.nproj * Kpad * 2);
    float *red = lds_red + MT * OB_DEC_WAVES * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gq = lane >> 4;
    const int G = gridDim.x;

    // tiles are numbered projection-major; this workgroup owns tiles b, b + G, b + 2G, ...
    int tile_base[4];
    tile_base[0] = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) tile_base[p + 1] = tile_base[p] + (p < A.nproj ? (A.p[p].N + 15) >> 4 : 0);
    const int ntiles = tile_base[3];
    const int nchunks = Kpad >> 9;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    const int per_tile = (nchunks - wave + OB_DEC_WAVES - 1) / OB_DEC_WAVES;   // this wave's chunks per tile

    // 1. issue every weight load of this wave: items (tile j, chunk wave + 8*ci)
    ob_u32x4 wreg[MT][PT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int t = blockIdx.x + j * G;
        const int p = t >= tile_base[2] ? 2 : (t >= tile_base[1] ? 1 : 0);
#pragma unroll
        for (int ci = 0; ci < PT; ++ci) {
            if (j < my_tiles && ci < per_tile)
                wreg[j][ci] = ob_dec_load_w(A.p[p], (t - tile_base[p]) << 4, wave + ci * OB_DEC_WAVES, lane);
            else
                wreg[j][ci] = (ob_u32x4){0u, 0u, 0u, 0u};
        }
    }

    // 2. prologue (statistics + activations into LDS) while the weights are in flight
    ob_dec_prologue(A, lds_a, red, Kpad);
    __syncthreads();

    // 3. MFMA
    ob_float4 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        acc[j] = (ob_float4){0.f, 0.f, 0.f, 0.f};
        const int t = blockIdx.x + j * G;
        const int p = t >= tile_base[2] ? 2 : (t >= tile_base[1] ? 1 : 0);
        const _Float16 *ap = lds_a + (size_t)p * Kpad;
#pragma unroll
        for (int ci = 0; ci < PT; ++ci) {
            if (j < my_tiles && ci < per_tile)
                ob_dec_chunk(wreg[j][ci], ap + (size_t)(wave + ci * OB_DEC_WAVES) * 512, gq, acc[j]);
        }
    }

    // 4. cross-wave reduction: column 0 of the result (lanes 0,16,32,48 hold rows 4*gq .. 4*gq+3)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        if (j < my_tiles && (lane & 15) == 0) {
            float *dst = lds_red + ((j * OB_DEC_WAVES + wave) << 4) + 4 * gq;
            dst[0] = acc[j][0]; dst[1] = acc[j][1]; dst[2] = acc[j][2]; dst[3] = acc[j][3];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < my_tiles * 16) {
        const int j = threadIdx.x >> 4, r = threadIdx.x & 15;
        float z = 0.f;
#pragma unroll
        for (int w = 0; w < OB_DEC_WAVES; ++w) z += lds_red[((j * OB_DEC_WAVES + w) << 4) + r];
        const int t = blockIdx.x + j * G;
        const int p = t >= tile_base[2] ? 2 : (t >= tile_base[1] ? 1 : 0);
        const int n = ((t - tile_base[p]) << 4) + r;
        if (n < A.p[p].N)       // z -> fp16 (bitnet.py:115), * g -> fp16 (:116)
            A.p[p].u[n] = (_Float16)(ob_round_h(z) * (float)A.p[p].g[n]);
    }
}

// ---------------------------------------------------------------------------------------------
// Attention for one new token (modeling_bitllama.py:522-563), one 256-thread workgroup per head.
// ---------------------------------------------------------------------------------------------
struct ObAttnArgs {
    const _Float16 *u_q, *u_k, *u_v;     // pre-LayerNorm projections [H*D], [Hkv*D], [Hkv*D]
    const _Float16 *cos, *sin;           // rope tables [max_pos, D] (fp16, as the reference caches them)
    _Float16 *kcache, *vcache;           // [Hkv, max_len, D]
    _Float16 *out;                       // [H*D]
    const int *pos;                      // device: position of the new token (= tokens already cached)
    int H, Hkv, D, max_len;
    float ln_eps;
};

__global__ __launch_bounds__(256) void ob_dec_attn_kernel(const ObAttnArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = A.D, H = A.H, Hkv = A.Hkv;
    const int head = blockIdx.x, kvh = head / (H / Hkv);
    const int pos = *A.pos;
    const int L = pos + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *red = reinterpret_cast<float *>(smem);                    // 6*16 floats
    _Float16 *q_s = reinterpret_cast<_Float16 *>(smem + 512);         // [D]
    _Float16 *k_s = q_s + D;                                         // [D] new key (post RoPE)
    _Float16 *v_s = k_s + D;                                         // [D] new value
    _Float16 *tmp = v_s + D;                                         // [2*D] pre-RoPE q, k
    float *sc = reinterpret_cast<float *>(tmp + 2 * D);              // [max_len] scores / probs
    float *po = sc + A.max_len;                                      // [4][D] partial outputs

    // LayerNorm statistics of the three rows (each workgroup recomputes them)
    const int NQ = H * D, NK = Hkv * D;
    const float cq = (float)A.u_q[0], ck = (float)A.u_k[0], cv = (float)A.u_v[0];
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int base = tid * 8; base < NQ; base += 256 * 8) {
        const ob_half8 t = *reinterpret_cast<const ob_half8 *>(A.u_q + base);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = (float)t[i] - cq; s[0] += d; s[1] += d * d; }
    }
    for (int base = tid * 8; base < NK; base += 256 * 8) {
        const ob_half8 tk = *reinterpret_cast<const ob_half8 *>(A.u_k + base);
        const ob_half8 tv = *reinterpret_cast<const ob_half8 *>(A.u_v + base);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float dk = (float)tk[i] - ck, dv = (float)tv[i] - cv;
            s[2] += dk; s[3] += dk * dk; s[4] += dv; s[5] += dv * dv;
        }
    }
    ob_block_sum_n<6>(s, red);
    float mq, rq, mk, rk, mv, rv;
    ob_ln_stats(s[0], s[1], cq, NQ, A.ln_eps, mq, rq);
    ob_ln_stats(s[2], s[3], ck, NK, A.ln_eps, mk, rk);
    ob_ln_stats(s[4], s[5], cv, NK, A.ln_eps, mv, rv);

    if (tid < D) {
        tmp[tid] = (_Float16)ob_ln_apply((float)A.u_q[head * D + tid], mq, rq);
        tmp[D + tid] = (_Float16)ob_ln_apply((float)A.u_k[kvh * D + tid], mk, rk);
        v_s[tid] = (_Float16)ob_ln_apply((float)A.u_v[kvh * D + tid], mv, rv);
    }
    __syncthreads();
    if (tid < D) {
        // apply_rotary_pos_emb (:175-181): q*cos + rotate_half(q)*sin, each op rounded to fp16
        const float c = (float)A.cos[(int64_t)pos * D + tid], sn = (float)A.sin[(int64_t)pos * D + tid];
        const int half = D >> 1;
        const float qr = tid < half ? -(float)tmp[tid + half] : (float)tmp[tid - half];
        const float kr = tid < half ? -(float)tmp[D + tid + half] : (float)tmp[D + tid - half];
        const float qe = ob_round_h(ob_round_h((float)tmp[tid] * c) + ob_round_h(qr * sn));
        const float ke = ob_round_h(ob_round_h((float)tmp[D + tid] * c) + ob_round_h(kr * sn));
        q_s[tid] = (_Float16)qe;
        k_s[tid] = (_Float16)ke;
        if (head % (H / Hkv) == 0) {      // one workgroup per kv head appends to the cache
            A.kcache[((int64_t)kvh * A.max_len + pos) * D + tid] = (_Float16)ke;
            A.vcache[((int64_t)kvh * A.max_len + pos) * D + tid] = v_s[tid];
        }
    }
    __syncthreads();

    // scores: one position per thread; fp32 dot -> fp16 (matmul output) -> / sqrt(D) -> fp16 (:546)
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const float sqrt_d = sqrtf((float)D);
    (void)inv_sqrt_d;
    const _Float16 *kbase = A.kcache + (int64_t)kvh * A.max_len * D;
    float lmax = -INFINITY;
    for (int p = tid; p < L; p += 256) {
        float dot = 0.f;
        if (p == pos) {
            for (int d = 0; d < D; d += 8) {
                const ob_half8 kk = *reinterpret_cast<const ob_half8 *>(k_s + d);
                const ob_half8 qq = *reinterpret_cast<const ob_half8 *>(q_s + d);
#pragma unroll
                for (int i = 0; i < 8; ++i) dot += (float)qq[i] * (float)kk[i];
            }
        } else {
            const _Float16 *kr = kbase + (int64_t)p * D;
            for (int d = 0; d < D; d += 8) {
                const ob_half8 kk = *reinterpret_cast<const ob_half8 *>(kr + d);
                const ob_half8 qq = *reinterpret_cast<const ob_half8 *>(q_s + d);
#pragma unroll
                for (int i = 0; i < 8; ++i) dot += (float)qq[i] * (float)kk[i];
            }
        }
        const float sv = ob_round_h(ob_round_h(dot) / sqrt_d);
        sc[p] = sv;
        lmax = fmaxf(lmax, sv);
    }
    // softmax in fp32 (:562), probabilities rounded to fp16
    lmax = ob_wave_max(lmax);
    __syncthreads();
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float ls[1] = {0.f};
    for (int p = tid; p < L; p += 256) {
        const float e = __expf(sc[p] - gmax);
        sc[p] = e;
        ls[0] += e;
    }
    ob_block_sum_n<1>(ls, red);
    const float inv_l = 1.0f / ls[0];
    for (int p = tid; p < L; p += 256) sc[p] = ob_round_h(sc[p] * inv_l);
    __syncthreads();

    // out = P . V: wave w takes positions p = w mod 4, lanes take dims (2 per lane up to D = 128)
    const _Float16 *vbase = A.vcache + (int64_t)kvh * A.max_len * D;
    for (int d0 = 0; d0 < D; d0 += 128) {
        const int d = d0 + 2 * lane;
        float o0 = 0.f, o1 = 0.f;
        if (d < D) {
            for (int p = wave; p < L; p += 4) {
                const float pr = sc[p];
                ob_half2 vv;
                if (p == pos) vv = *reinterpret_cast<const ob_half2 *>(v_s + d);
                else vv = *reinterpret_cast<const ob_half2 *>(vbase + (int64_t)p * D + d);
                o0 += pr * (float)vv[0];
                o1 += pr * (float)vv[1];
            }
            po[wave * D + d] = o0;
            po[wave * D + d + 1] = o1;
        }
    }
    __syncthreads();
    if (tid < D) {
        const float o = po[tid] + po[D + tid] + po[2 * D + tid] + po[3 * D + tid];
        A.out[head * D + tid] = (_Float16)o;
    }
}

// ---------------------------------------------------------------------------------------------
// Final norm + fp16 lm_head GEMV + per-workgroup argmax (modeling_bitllama.py:1321,1610-1611;
// generation/utils.py:2540).  Persistent grid, one row per wave iteration.
// ---------------------------------------------------------------------------------------------
struct ObHeadArgs {
    const _Float16 *hres_in, *u_prev, *rms_w;   // residual stream, last down_proj pre-LN, final norm weight
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
        ob_block_sum_n<2>(s, red);
        float mean, rstd;
        ob_ln_stats(s[0], s[1], c, K, A.ln_eps, mean, rstd);
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
        ob_block_sum_n<1>(ss, red);
        const float rs = rsqrtf(ss[0] / (float)K + A.rms_eps);
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
    // rows: wave-strided over the vocabulary; 8 halves per lane per 512-wide step
    float best = -INFINITY;
    int besti = 0x7fffffff;
    const int gw = blockIdx.x * OB_DEC_WAVES + wave, nw = gridDim.x * OB_DEC_WAVES;
    for (int row = gw; row < A.V; row += nw) {
        const _Float16 *wr = A.lm_w + (int64_t)row * K;
        float acc = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            const ob_half8 wv = __builtin_nontemporal_load(reinterpret_cast<const ob_half8 *>(wr + k));
            const ob_half8 xv = *reinterpret_cast<const ob_half8 *>(x_s + k);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += (float)wv[i] * (float)xv[i];
        }
        acc = ob_wave_sum(acc);
        const float lg = ob_round_h(acc);
        if (lane == 0) A.logits[row] = (_Float16)lg;
        if (lg > best || (lg == best && row < besti)) { best = lg; besti = row; }
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
                                                            int *token, int *pos, int *out_tokens, int max_out)
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
        *token = si[0];
        if (out_tokens && p < max_out) out_tokens[p] = si[0];
        *pos = p + 1;
    }
}
