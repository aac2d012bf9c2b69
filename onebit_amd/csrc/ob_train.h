// Train-mode 1-bit linear layer: forward and backward of the reference's
// BitLinear + SignSTE (transformers/src/transformers/models/bitnet.py:14-28, 58-68) on LATENT full-precision
// weights, for knowledge-distillation training on MI355X.
//
//   forward   a = x * h;  S = sign(W) (sign(0) = 0, :18);  z = a . S^T;  u = z * g;  y = LayerNorm(u) (+ bias)
//   backward  gu = LayerNorm'(gy);  gz = gu * g;  gg = sum_t gu * z;  gbias = sum_t gy
//             ga = gz . S;  gx = ga * h;  gh = sum_t ga * x
//             gS = gz^T . a;  gW = gS * (1.001 - tanh(W)^2)                       (the STE of :21-23)
//
// Three GEMMs share one MFMA tile kernel whose operand loaders apply the elementwise ops on the way into LDS
// (x * h, sign(W) -- the sign matrix is never materialised in HBM) and whose epilogue applies what follows
// (* h for gx, the STE factor for gW).  fp16 tensors run v_mfma_f32_16x16x16_f16, fp32 tensors
// v_mfma_f32_16x16x4_f32; accumulation is fp32 in both.  In fp16 every tensor-level op of the reference rounds
// once to fp16 (x * h, z, u, gz, ga, gx, gW), as the torch ops do.
#pragma once
#include "ob_common.h"

enum { OB_TX_NONE = 0, OB_TX_SCALE_R = 1, OB_TX_SCALE_I = 2, OB_TX_SIGN = 3 };
enum { OB_TE_PLAIN = 0, OB_TE_GX = 1, OB_TE_STE = 2 };

#define OB_TG_BM 64
#define OB_TG_BN 64
#define OB_TG_BR 32
#define OB_TG_THREADS 256

struct ObTGemmArgs {
    const void *A, *B;        // C[m, n] = sum_r A(m, r) * B(n, r)
    long long lda, ldb;       // leading dimension of the operand's storage (elements)
    const void *va, *vb;      // transform vectors (same dtype), or null
    void *C;                  // [M, N] row-major, pitch ldc
    void *C2;                 // OB_TE_GX: C2 = C * vc[n]
    const void *vc;           // OB_TE_GX: vector [N];  OB_TE_STE: latent W [M, N] (pitch ldc)
    long long ldc;
    int M, N, R;
};

template <typename TI> struct ObTgT;
template <> struct ObTgT<_Float16> { static constexpr int PITCH = OB_TG_BR + 8; };     // 80-byte rows
template <> struct ObTgT<float> { static constexpr int PITCH = OB_TG_BR + 4; };        // 144-byte rows

template <typename TI>
__device__ __forceinline__ TI ob_tg_sign(TI v)
{
    return (TI)((v > (TI)0) ? 1.0f : ((v < (TI)0) ? -1.0f : 0.0f));      // NaN -> 0, as (v > 0) - (v < 0)
}

// One 64 x 32 operand tile into LDS as S[i][r].  RC: the reduction index r is the contiguous one in memory
// (element (i, r) at P[i * ld + r]); otherwise the output index i is (element at P[r * ld + i]) and the tile is
// transposed on its way in.  8 contiguous elements per thread, one 16-byte (fp16) or two 16-byte (fp32) loads
// when the run is inside the matrix and aligned.
template <typename TI, bool RC, int TX>
__device__ __forceinline__ void ob_tg_load(TI (*S)[ObTgT<TI>::PITCH], const TI *__restrict__ P, long long ld,
                                           const TI *__restrict__ vec, int i0, int r0, int I, int R, int tid)
{
    TI v[8];
    int gi, gr;
    if (RC) { gi = i0 + (tid >> 2); gr = r0 + (tid & 3) * 8; }
    else { gr = r0 + (tid >> 3); gi = i0 + (tid & 7) * 8; }
    const TI *p = RC ? P + (long long)gi * ld + gr : P + (long long)gr * ld + gi;
    const bool inside = RC ? (gi < I && gr + 8 <= R) : (gr < R && gi + 8 <= I);
    if (inside && ((uintptr_t)p % 16) == 0) {
        if (sizeof(TI) == 2) {
            *reinterpret_cast<ob_u32x4 *>(v) = *reinterpret_cast<const ob_u32x4 *>(p);
        } else {
            reinterpret_cast<ob_u32x4 *>(v)[0] = reinterpret_cast<const ob_u32x4 *>(p)[0];
            reinterpret_cast<ob_u32x4 *>(v)[1] = reinterpret_cast<const ob_u32x4 *>(p)[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = RC ? (gi < I && gr + e < R) : (gr < R && gi + e < I);
            v[e] = ok ? p[e] : (TI)0;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ei = RC ? gi : gi + e, er = RC ? gr + e : gr;
        if (TX == OB_TX_SIGN) v[e] = ob_tg_sign<TI>(v[e]);
        if (TX == OB_TX_SCALE_R) v[e] = (ei < I && er < R) ? (TI)(v[e] * vec[er]) : (TI)0;
        if (TX == OB_TX_SCALE_I) v[e] = (ei < I && er < R) ? (TI)(v[e] * vec[ei]) : (TI)0;
    }
    if (RC) {
        TI *d = &S[tid >> 2][(tid & 3) * 8];
        if (sizeof(TI) == 2) *reinterpret_cast<ob_u32x4 *>(d) = *reinterpret_cast<ob_u32x4 *>(v);
        else { reinterpret_cast<ob_u32x4 *>(d)[0] = reinterpret_cast<ob_u32x4 *>(v)[0]; reinterpret_cast<ob_u32x4 *>(d)[1] = reinterpret_cast<ob_u32x4 *>(v)[1]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) S[(tid & 7) * 8 + e][tid >> 3] = v[e];
    }
}

template <typename TI, bool RCA, bool RCB, int TXA, int TXB, int EPI>
__global__ __launch_bounds__(OB_TG_THREADS) void ob_tgemm_kernel(const ObTGemmArgs a)
{
    constexpr int PITCH = ObTgT<TI>::PITCH;
    __shared__ __attribute__((aligned(16))) TI As[OB_TG_BM][PITCH];
    __shared__ __attribute__((aligned(16))) TI Bs[OB_TG_BN][PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int i0 = blockIdx.y * OB_TG_BM, j0 = blockIdx.x * OB_TG_BN;
    const int lr = lane & 15, lg = lane >> 4;
    ob_float4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (ob_float4){0.f, 0.f, 0.f, 0.f};
    const TI *A = (const TI *)a.A, *B = (const TI *)a.B;
    for (int r0 = 0; r0 < a.R; r0 += OB_TG_BR) {
        ob_tg_load<TI, RCA, TXA>(As, A, a.lda, (const TI *)a.va, i0, r0, a.M, a.R, tid);
        ob_tg_load<TI, RCB, TXB>(Bs, B, a.ldb, (const TI *)a.vb, j0, r0, a.N, a.R, tid);
        __syncthreads();
        if (sizeof(TI) == 2) {
#pragma unroll
            for (int kk = 0; kk < OB_TG_BR; kk += 16) {
                ob_half4 fa[2], fb[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    fa[x] = *reinterpret_cast<const ob_half4 *>(&As[wm * 32 + x * 16 + lr][kk + 4 * lg]);
                    fb[x] = *reinterpret_cast<const ob_half4 *>(&Bs[wn * 32 + x * 16 + lr][kk + 4 * lg]);
                }
                // the N side is the MFMA's row operand: a lane ends up with 4 consecutive n of one m
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x16f16(fb[y], fa[x], acc[x][y], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < OB_TG_BR; kk += 4) {
                float fa[2], fb[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    fa[x] = (float)As[wm * 32 + x * 16 + lr][kk + lg];
                    fb[x] = (float)Bs[wn * 32 + x * 16 + lr][kk + lg];
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[y], fa[x], acc[x][y], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    TI *C = (TI *)a.C, *C2 = (TI *)a.C2;
    const TI *vc = (const TI *)a.vc;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int m = i0 + wm * 32 + x * 16 + lr;
        if (m >= a.M) continue;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int n = j0 + wn * 32 + y * 16 + 4 * lg;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= a.N) continue;
                const long long o = (long long)m * a.ldc + n + e;
                const TI c = (TI)acc[x][y][e];
                if (EPI == OB_TE_PLAIN) C[o] = c;
                if (EPI == OB_TE_GX) { C[o] = c; C2[o] = (TI)(c * vc[n + e]); }
                if (EPI == OB_TE_STE) {
                    // SignSTEFunc.backward (bitnet.py:22-23): grad * (1.001 - tanh(w) ** 2) as TENSOR ops in the tensor
                    // dtype -- tanh, the square, the subtraction (the scalar 1.001 becomes TI: 1.000977 in fp16) and the
                    // product each round once to TI
                    const TI t = (TI)tanhf((float)vc[o]);
                    const TI t2 = (TI)((float)t * (float)t);
                    const TI f = (TI)((float)(TI)1.001f - (float)t2);
                    C[o] = (TI)((float)c * (float)f);
                }
            }
        }
    }
}

// ---- LayerNorm of the train-mode layer, one workgroup per token row ---------------------------------------------
// forward: u = z * g (rounded to the tensor dtype, bitnet.py:64), y = (u - mean) * rstd (+ bias) (:66-68), biased
// variance, statistics in fp32 (two-pass); stats[t] = {mean, rstd} for the backward pass.
template <typename TI>
__global__ __launch_bounds__(256) void ob_train_ln_fwd_kernel(const TI *__restrict__ z, const TI *__restrict__ g, const TI *__restrict__ bias,
                                                              TI *__restrict__ y, float *__restrict__ stats, int N, float eps)
{
    __shared__ float red[8];
    const int t = blockIdx.x;
    const TI *zr = z + (long long)t * N;
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) s += (float)(TI)(zr[n] * g[n]);
    const float mean = ob_block_sum(s, red) / (float)N;
    float q = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) { const float d = (float)(TI)(zr[n] * g[n]) - mean; q += d * d; }
    const float var = ob_block_sum(q, red) / (float)N;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int n = threadIdx.x; n < N; n += 256) {
        TI o = (TI)(((float)(TI)(zr[n] * g[n]) - mean) * rstd);
        if (bias) o = (TI)(o + bias[n]);
        y[(long long)t * N + n] = o;
    }
    if (threadIdx.x == 0) { stats[2 * t] = mean; stats[2 * t + 1] = rstd; }
}

// backward through the LayerNorm and the * g: with uh = (u - mean) * rstd,
//   gu = rstd * (gy - mean_n(gy) - uh * mean_n(gy * uh));   gz = gu * g
// rowc[t] = {mean_n(gy), mean_n(gy * uh)} is kept for the column reductions (gg needs gu again).
template <typename TI>
__global__ __launch_bounds__(256) void ob_train_ln_bwd_kernel(const TI *__restrict__ gy, const TI *__restrict__ z, const TI *__restrict__ g,
                                                              const float *__restrict__ stats, TI *__restrict__ gz, float *__restrict__ rowc, int N)
{
    __shared__ float red[8];
    const int t = blockIdx.x;
    const TI *zr = z + (long long)t * N, *gr = gy + (long long)t * N;
    const float mean = stats[2 * t], rstd = stats[2 * t + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float uh = ((float)(TI)(zr[n] * g[n]) - mean) * rstd, d = (float)gr[n];
        s1 += d; s2 += d * uh;
    }
    const float c1 = ob_block_sum(s1, red) / (float)N;
    const float c2 = ob_block_sum(s2, red) / (float)N;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float uh = ((float)(TI)(zr[n] * g[n]) - mean) * rstd;
        const TI gu = (TI)(rstd * ((float)gr[n] - c1 - uh * c2));
        gz[(long long)t * N + n] = (TI)(gu * g[n]);
    }
    if (threadIdx.x == 0) { rowc[2 * t] = c1; rowc[2 * t + 1] = c2; }
}

// column reductions over the T rows (deterministic: fixed assignment and order).  64 columns per workgroup, 4 row
// lanes; gg[n] = sum_t gu[t, n] * z[t, n], gbias[n] = sum_t gy[t, n].
template <typename TI>
__global__ __launch_bounds__(256) void ob_train_cols_ln_kernel(const TI *__restrict__ gy, const TI *__restrict__ z, const TI *__restrict__ g,
                                                               const float *__restrict__ stats, const float *__restrict__ rowc,
                                                               TI *__restrict__ gg, TI *__restrict__ gbias, int T, int N)
{
    __shared__ float sm[2][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
    float a0 = 0.f, a1 = 0.f;
    if (n < N) {
        const float gn = (float)g[n];
        for (int t = rl; t < T; t += 4) {
            const float mean = stats[2 * t], rstd = stats[2 * t + 1], c1 = rowc[2 * t], c2 = rowc[2 * t + 1];
            const TI zz = z[(long long)t * N + n];
            const float uh = ((float)(TI)(zz * (TI)gn) - mean) * rstd, d = (float)gy[(long long)t * N + n];
            const TI gu = (TI)(rstd * (d - c1 - uh * c2));
            a0 += (float)gu * (float)zz;
            a1 += d;
        }
    }
    sm[0][rl][c] = a0; sm[1][rl][c] = a1;
    __syncthreads();
    if (rl == 0 && n < N) {
        gg[n] = (TI)((sm[0][0][c] + sm[0][1][c]) + (sm[0][2][c] + sm[0][3][c]));
        if (gbias) gbias[n] = (TI)((sm[1][0][c] + sm[1][1][c]) + (sm[1][2][c] + sm[1][3][c]));
    }
}

// gh[k] = sum_t ga[t, k] * x[t, k]
template <typename TI>
__global__ __launch_bounds__(256) void ob_train_cols_gh_kernel(const TI *__restrict__ ga, const TI *__restrict__ x, TI *__restrict__ gh, int T, int K)
{
    __shared__ float sm[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, k = blockIdx.x * 64 + c;
    float a0 = 0.f;
    if (k < K)
        for (int t = rl; t < T; t += 4) a0 += (float)(TI)(ga[(long long)t * K + k] * x[(long long)t * K + k]);
    sm[rl][c] = a0;
    __syncthreads();
    if (rl == 0 && k < K) gh[k] = (TI)((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]));
}
