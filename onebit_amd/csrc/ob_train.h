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

// ---- fp16 on gfx950: 128 x 128 x 32 tiles, v_mfma_f32_16x16x32_f16, double-buffered LDS ---------------------------------
// Same operands, transforms, rounding points and epilogues as ob_tgemm_kernel<_Float16, ...> (which stays the fp32 path and the
// definition of the arithmetic); what changes is the tiling and the way operands travel:
//   * 4 waves as 2 x 2, a wave owns 64 x 64 of the output: 16 MFMAs per 8 fragments
//   * the next tile's operands are fetched into registers before this tile's MFMAs and written -- transform applied -- to the
//     other LDS buffer after them: one barrier per 32-deep step
//   * an operand whose OUTPUT index is the contiguous one in memory (sign(W) in ga = gz . S, both operands of gS = gz^T . a) is
//     no longer transposed element by element on its way into LDS (8 two-byte stores per thread): it is stored as it lies,
//     S[r][i] in 16-byte pieces, XOR-swizzled, and the MFMA fragment (8 consecutive r of one i) comes out of two
//     ds_read_b64_tr_b16 -- the V-operand recipe of ob_flash.h
// Layer figure (tools/train_probe.py / bench.py train_layer: T = 4096, 4096 -> 11008, forward + backward, 3 GEMMs): 134 TFLOP/s with the
// round-3 kernel, 249 with this tiling, 338 with the scale transform as packed multiplies (it was 8 bounds-checked scalar products per
// thread and operand), 418 with the two-stage column sums below; the three GEMMs alone run at 430-510 TFLOP/s.
#define OB_TG2_B 128
#ifndef OB_TG2_BR
#define OB_TG2_BR 32                                     // reduction depth of a step (32, or 64: measured equal -- 2.70 vs 2.65 ms per layer step)
#endif
#define OB_TG2_PITCH (OB_TG2_BR + 8)                     // RC image: S[i][r]
#define OB_TG2_TILE (OB_TG2_B * OB_TG2_PITCH)            // halves per operand buffer (the transposed image [BR][128] is smaller)
#define OB_TG2_LDS (4 * OB_TG2_TILE * 2)                 // two operands x two buffers, bytes (dynamic: 72 KB at depth 64)

// where this thread's 8 contiguous elements of the (half-)tile sit: RC (r contiguous) i = tid >> 2, r = 8 (tid & 3);
// otherwise r = tid >> 3, i = 8 (tid & 7)
// (j = which 32 of the step's reduction indices)
template <bool RC>
__device__ __forceinline__ ob_u32x4 ob_tg2_fetch(const _Float16 *__restrict__ P, long long ld, int i0, int r0, int I, int R, int tid, int j)
{
    int gi, gr;
    if (RC) { gi = i0 + (tid >> 2); gr = r0 + 32 * j + (tid & 3) * 8; }
    else { gr = r0 + 32 * j + (tid >> 3); gi = i0 + (tid & 7) * 8; }
    const _Float16 *p = RC ? P + (long long)gi * ld + gr : P + (long long)gr * ld + gi;
    const bool inside = RC ? (gi < I && gr + 8 <= R) : (gr < R && gi + 8 <= I);
    if (inside && ((uintptr_t)p % 16) == 0) return *reinterpret_cast<const ob_u32x4 *>(p);
    ob_half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool ok = RC ? (gi < I && gr + e < R) : (gr < R && gi + e < I);
        v[e] = ok ? p[e] : (_Float16)0;
    }
    return __builtin_bit_cast(ob_u32x4, v);
}
// the swizzle of the transposed image: piece pc (8 halves) of row r sits at pc ^ 2 f(r), f(r) = (r & 3) | ((r >> 3) & 1) << 2 -- the
// 8 rows a half-wave of the transpose read touches (rows 8 lg + q of two lane groups, 32 bytes each) then fall on disjoint banks
__device__ __forceinline__ int ob_tg2_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }
// ... transformed and written to LDS: S[i][r] (RC) or S[r][i] swizzled; `hf` = which 64 of the tile's 128 output indices
template <bool RC, int TX>
__device__ __forceinline__ void ob_tg2_put(_Float16 *S, int hf, ob_u32x4 raw, const _Float16 *__restrict__ vec, int i0, int r0, int I, int R, int tid, int j)
{
    ob_half8 v = __builtin_bit_cast(ob_half8, raw);
    int gi, gr;
    if (RC) { gi = i0 + (tid >> 2); gr = r0 + 32 * j + (tid & 3) * 8; }
    else { gr = r0 + 32 * j + (tid >> 3); gi = i0 + (tid & 7) * 8; }
    if (TX == OB_TX_SIGN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ob_tg_sign<_Float16>(v[e]);
    }
    if (TX == OB_TX_SCALE_R || TX == OB_TX_SCALE_I) {
        // elements outside the matrix were fetched as zeros: the factor only has to be a finite number there.  The factor vector
        // runs along the thread's 8 elements (one 16-byte load, four packed multiplies) or is one scalar for all of them.
        const bool along = (TX == OB_TX_SCALE_R) == RC;
        const int q0 = TX == OB_TX_SCALE_R ? gr : gi, Q = TX == OB_TX_SCALE_R ? R : I;
        ob_half8 f;
        if (along) {
            if (q0 + 8 <= Q && ((uintptr_t)(vec + q0) % 16) == 0) f = *reinterpret_cast<const ob_half8 *>(vec + q0);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = q0 + e < Q ? vec[q0 + e] : (_Float16)0;
            }
        } else {
            const _Float16 f1 = q0 < Q ? vec[q0] : (_Float16)0;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f1;
        }
        v = v * f;                                          // fp16 products, rounded once (bitnet.py:62 / the STE operand)
    }
    if (RC) {
        *reinterpret_cast<ob_half8 *>(S + (64 * hf + (tid >> 2)) * OB_TG2_PITCH + 32 * j + (tid & 3) * 8) = v;
    } else {
        const int r = 32 * j + (tid >> 3), pc = 8 * hf + (tid & 7);
        *reinterpret_cast<ob_half8 *>(S + r * OB_TG2_B + 8 * (pc ^ ob_tg2_swz(r))) = v;
    }
}
// MFMA fragment of output index ib + lr: the 8 reduction elements 8 lg .. 8 lg + 7
template <bool RC>
__device__ __forceinline__ ob_half8 ob_tg2_frag(const _Float16 *S, int ib, int lr, int lg, int kk)
{
    if (RC) return *reinterpret_cast<const ob_half8 *>(S + (ib + lr) * OB_TG2_PITCH + 32 * kk + 8 * lg);
    typedef short ob_v4s __attribute__((ext_vector_type(4)));
    typedef short ob_v8s __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) ob_v4s ob_lds_v4s;
    // this lane's chunk of the 4-row x 16-column block the transpose read gathers: row lr >> 2, columns 4 (lr & 3) .. + 3
    const int row = 32 * kk + 8 * lg + (lr >> 2), pc = (ib >> 3) + ((lr & 3) >> 1);
    const _Float16 *p = S + row * OB_TG2_B + 8 * (pc ^ ob_tg2_swz(row)) + 4 * (lr & 1);      // (rows row and row + 4 share the swizzle)
    const ob_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)p);
    const ob_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(p + 4 * OB_TG2_B));
    const ob_v8s a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(ob_half8, a8);
}

template <bool RCA, bool RCB, int TXA, int TXB, int EPI>
__global__ __launch_bounds__(OB_TG_THREADS, 2) void ob_tgemm128_f16_kernel(const ObTGemmArgs a)
{
    typedef _Float16 TI;
    constexpr int NJ = OB_TG2_BR / 32;
    extern __shared__ __attribute__((aligned(16))) TI ob_tg2_smem[];
    TI *const As = ob_tg2_smem, *const Bs = ob_tg2_smem + 2 * OB_TG2_TILE;     // [2][OB_TG2_TILE] each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int i0 = blockIdx.y * OB_TG2_B, j0 = blockIdx.x * OB_TG2_B;
    const int lr = lane & 15, lg = lane >> 4;
    ob_float4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = (ob_float4){0.f, 0.f, 0.f, 0.f};
    const TI *A = (const TI *)a.A, *B = (const TI *)a.B, *va = (const TI *)a.va, *vb = (const TI *)a.vb;
    ob_u32x4 ra[2][NJ], rb[2][NJ];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                ra[hf][j] = ob_tg2_fetch<RCA>(A, a.lda, i0 + 64 * hf, r0, a.M, a.R, tid, j);
                rb[hf][j] = ob_tg2_fetch<RCB>(B, a.ldb, j0 + 64 * hf, r0, a.N, a.R, tid, j);
            }
    };
    auto put = [&](int buf, int r0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                ob_tg2_put<RCA, TXA>(As + buf * OB_TG2_TILE, hf, ra[hf][j], va, i0 + 64 * hf, r0, a.M, a.R, tid, j);
                ob_tg2_put<RCB, TXB>(Bs + buf * OB_TG2_TILE, hf, rb[hf][j], vb, j0 + 64 * hf, r0, a.N, a.R, tid, j);
            }
    };
    fetch(0);
    put(0, 0);
    __syncthreads();
    int buf = 0;
    for (int r0 = 0; r0 < a.R; r0 += OB_TG2_BR, buf ^= 1) {
        const bool more = r0 + OB_TG2_BR < a.R;
        if (more) fetch(r0 + OB_TG2_BR);                // in flight underneath this step's MFMAs
#pragma unroll
        for (int kk = 0; kk < NJ; ++kk) {
            ob_half8 fa[4], fb[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                fa[x] = ob_tg2_frag<RCA>(As + buf * OB_TG2_TILE, wm * 64 + x * 16, lr, lg, kk);
                fb[x] = ob_tg2_frag<RCB>(Bs + buf * OB_TG2_TILE, wn * 64 + x * 16, lr, lg, kk);
            }
            // the N side is the MFMA's row operand: a lane ends up with 4 consecutive n of one m
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[y], fa[x], acc[x][y], 0, 0, 0);
        }
        if (more) put(buf ^ 1, r0 + OB_TG2_BR);         // (its last readers passed the barrier that ended the previous step)
        __syncthreads();
    }
    TI *C = (TI *)a.C, *C2 = (TI *)a.C2;
    const TI *vc = (const TI *)a.vc;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int m = i0 + wm * 64 + x * 16 + lr;
        if (m >= a.M) continue;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int n = j0 + wn * 64 + y * 16 + 4 * lg;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= a.N) continue;
                const long long o = (long long)m * a.ldc + n + e;
                const TI c = (TI)acc[x][y][e];
                if (EPI == OB_TE_PLAIN) C[o] = c;
                if (EPI == OB_TE_GX) { C[o] = c; C2[o] = (TI)(c * vc[n + e]); }
                if (EPI == OB_TE_STE) {                 // (see ob_tgemm_kernel)
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

// Column sums over the tokens, in two deterministic stages (round 4; one 64-column block walking all T rows took 0.45 / 0.30 ms of a
// 3.3 ms layer step): stage 1 -- block (column block, token slice s of OB_TC_SLICES) accumulates its rows in fp32 and writes
// part[s][column]; stage 2 -- one thread per column adds the slices in order and rounds once to the tensor dtype.
//   gg[n] = sum_t gu[t, n] * z[t, n], gbias[n] = sum_t gy[t, n]      (gu: the LayerNorm backward of ob_train_ln_bwd_kernel, recomputed)
//   gh[k] = sum_t ga[t, k] * x[t, k]
#define OB_TC_SLICES 32
template <typename TI>
__global__ __launch_bounds__(256) void ob_train_cols_ln_kernel(const TI *__restrict__ gy, const TI *__restrict__ z, const TI *__restrict__ g,
                                                               const float *__restrict__ stats, const float *__restrict__ rowc,
                                                               float *__restrict__ part, int T, int N)
{
    __shared__ float sm[2][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
    const int per = (T + OB_TC_SLICES - 1) / OB_TC_SLICES, t0 = blockIdx.y * per, t1 = min(T, t0 + per);
    float a0 = 0.f, a1 = 0.f;
    if (n < N) {
        const float gn = (float)g[n];
        for (int t = t0 + rl; t < t1; t += 4) {
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
        part[((size_t)blockIdx.y * 2 + 0) * N + n] = (sm[0][0][c] + sm[0][1][c]) + (sm[0][2][c] + sm[0][3][c]);
        part[((size_t)blockIdx.y * 2 + 1) * N + n] = (sm[1][0][c] + sm[1][1][c]) + (sm[1][2][c] + sm[1][3][c]);
    }
}

template <typename TI>
__global__ __launch_bounds__(256) void ob_train_cols_gh_kernel(const TI *__restrict__ ga, const TI *__restrict__ x, float *__restrict__ part, int T, int K)
{
    __shared__ float sm[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, k = blockIdx.x * 64 + c;
    const int per = (T + OB_TC_SLICES - 1) / OB_TC_SLICES, t0 = blockIdx.y * per, t1 = min(T, t0 + per);
    float a0 = 0.f;
    if (k < K)
        for (int t = t0 + rl; t < t1; t += 4) a0 += (float)(TI)(ga[(long long)t * K + k] * x[(long long)t * K + k]);
    sm[rl][c] = a0;
    __syncthreads();
    if (rl == 0 && k < K) part[(size_t)blockIdx.y * K + k] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
}

// stage 2: out_j[c] = TI(sum over slices of part[s][j][c]), j < NOUT (out_1 may be null)
template <typename TI, int NOUT>
__global__ __launch_bounds__(256) void ob_train_cols_finish_kernel(const float *__restrict__ part, TI *__restrict__ out0, TI *__restrict__ out1, int C)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        float a = 0.f;
        for (int sl = 0; sl < OB_TC_SLICES; ++sl) a += part[((size_t)sl * NOUT + j) * C + c];
        TI *o = j == 0 ? out0 : out1;
        if (o) o[c] = (TI)a;
    }
}
