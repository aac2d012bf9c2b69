/*
 * onebit_oracle.c -- CPU restatement of OneBit's packed 1-bit linear layer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path in
 * onebit_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it; the product path never does.
 *
 * Every function restates one piece of the reference (paths relative to
 * /root/reference):
 *   scripts/convert_llama_to_infer_ckpt.py:7-15   fp16_to_int8   (sign packer)
 *   transformers/src/transformers/models/bitnet.py:98-110  int8_to_fp16 (unpack)
 *   transformers/src/transformers/models/bitnet.py:112-122 BitLinearInf.forward
 *
 * The reference has no tests or golden vectors of its own, so
 * this oracle is pinned by fixtures generated in the build container by
 * importing the reference module itself: tests/golden/gen_goldens.py writes
 * the .npz fixtures under tests/golden/ and tests/test_oracle_golden.py checks this file against
 * them.
 *
 * Plain C99, no dependencies; fp16 is emulated with explicit
 * round-to-nearest-even conversions so results do not depend on compiler
 * _Float16 support.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ fp16 -- */

static inline float ob_u32_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t ob_f32_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* IEEE binary16 -> binary32 (exact). */
float ob_half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return ob_u32_as_f32(sign);
        /* subnormal: man * 2^-24 */
        float f = (float)man * (1.0f / 16777216.0f);
        return (sign ? -f : f);
    }
    if (exp == 31) return ob_u32_as_f32(sign | 0x7f800000u | (man << 13));
    return ob_u32_as_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* binary32 -> binary16, round to nearest even (what torch .to(float16) and the
 * GPU v_cvt_f16_f32 do in the default rounding mode). */
uint16_t ob_float_to_half(float f)
{
    uint32_t x = ob_f32_as_u32(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) {                       /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u | ((ax >> 13) & 0x3ffu) : 0));
    }
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  /* >= 65520 -> inf */
    if (ax < 0x38800000u) {                        /* < 2^-14: subnormal or zero */
        if (ax < 0x33000000u) return (uint16_t)sign;   /* < 2^-25 -> 0 (2^-25 ties to even 0) */
        /* value = m * 2^(e-150), want round(value / 2^-24) */
        uint32_t e = ax >> 23;
        uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126u - e;                 /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    {
        uint32_t e = (ax >> 23) - 112u;
        uint32_t m = ax & 0x7fffffu;
        uint32_t r = (e << 10) | (m >> 13);
        uint32_t rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;   /* may carry into exponent: correct */
        return (uint16_t)(sign | r);
    }
}

static inline float ob_round_h(float f) { return ob_half_to_float(ob_float_to_half(f)); }

/* ------------------------------------------------------------- packing -- */

/*
 * fp16_to_int8 (convert_llama_to_infer_ckpt.py:7-15) on a float tensor s[N,K]
 * that holds sign values (+1, -1, or 0 as produced by torch.sign):
 *   bit   = uint8((0 - s + 1) / 2)      -> +1 -> 0, 0 -> 0 (0.5 truncates), -1 -> 1
 *   byte  = sum_i bit[8j+i] << i        (LSB first, :12-13), stored as int8
 * Returns 0, or -1 if K % 8 != 0 (the reference's view(...,-1,8) raises).
 */
int ob_oracle_fp16_to_int8(const float *s, int8_t *out, int64_t N, int64_t K)
{
    if (K % 8 != 0 || N < 0 || K < 0) return -1;
    for (int64_t n = 0; n < N; n++) {
        for (int64_t j = 0; j < K / 8; j++) {
            unsigned byte = 0;
            for (int i = 0; i < 8; i++) {
                float v = (0.0f - s[n * K + 8 * j + i] + 1.0f) / 2.0f;
                /* .to(torch.uint8) truncates toward zero; NaN -> 0 */
                unsigned bit = (v >= 1.0f) ? (unsigned)v : 0u;
                byte += (bit << i);              /* uint8 matmul wraps mod 256 */
            }
            out[n * (K / 8) + j] = (int8_t)(uint8_t)(byte & 0xffu);
        }
    }
    return 0;
}

/*
 * Converter main loop body (convert_llama_to_infer_ckpt.py:29-32): packed =
 * fp16_to_int8(torch.sign(w)) for latent weights w[N,K].  sign(0) = 0 -> bit 0
 * (= +1); NaN -> bit 0.
 */
int ob_oracle_pack_signs(const float *w, int8_t *out, int64_t N, int64_t K)
{
    if (K % 8 != 0 || N < 0 || K < 0) return -1;
    for (int64_t n = 0; n < N; n++) {
        for (int64_t j = 0; j < K / 8; j++) {
            unsigned byte = 0;
            for (int i = 0; i < 8; i++)
                byte |= (unsigned)(w[n * K + 8 * j + i] < 0.0f) << i;
            out[n * (K / 8) + j] = (int8_t)(uint8_t)byte;
        }
    }
    return 0;
}

/*
 * int8_to_fp16 (bitnet.py:98-110): bit_k = (byte >> (k % 8)) & 1 with an
 * arithmetic shift on the signed byte (bit 7 still comes out right),
 * value = -2*bit + 1.  Output is float (+1.0 / -1.0), exactly representable in
 * fp16 and fp32 alike.
 */
void ob_oracle_unpack(const int8_t *packed, float *out, int64_t N, int64_t K)
{
    for (int64_t n = 0; n < N; n++)
        for (int64_t k = 0; k < K; k++) {
            int v = (int)packed[n * (K / 8) + k / 8];      /* sign-extended */
            int bit = (v >> (k % 8)) & 1;
            out[n * K + k] = (float)(-2 * bit + 1);
        }
}

/* -------------------------------------------------------------- forward -- */

/*
 * BitLinearInf.forward (bitnet.py:112-122), fp32 parameters and input:
 *   a = x * h                         (:113)   fp32
 *   z = a . W^T                       (:115)   fp32 GEMM; here accumulated in fp64
 *   u = z * g                         (:116)
 *   y = LayerNorm(u), eps, biased var (:118)   stats in fp64 here
 *   y += bias                         (:119-120)
 * x [T,K], h [K], g [N], bias [N] or NULL, packed [N,K/8].
 * y_out [T,N] (post-LN), u_out [T,N] or NULL (pre-LN, after *g).
 */
int ob_oracle_forward_f32(const int8_t *packed, const float *x, const float *h,
                          const float *g, const float *bias, float *y_out,
                          float *u_out, int64_t T, int64_t K, int64_t N, float eps)
{
    if (K % 8 != 0) return -1;
    float *a = (float *)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    double *u = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    if (!a || !u) { free(a); free(u); return -2; }
    for (int64_t t = 0; t < T; t++) {
        for (int64_t k = 0; k < K; k++) a[k] = x[t * K + k] * h[k];
        for (int64_t n = 0; n < N; n++) {
            const uint8_t *row = (const uint8_t *)packed + n * (K / 8);
            double acc = 0.0;
            for (int64_t k = 0; k < K; k++)
                acc += ((row[k >> 3] >> (k & 7)) & 1) ? -(double)a[k] : (double)a[k];
            float z = (float)acc;
            u[n] = (double)(z * g[n]);
        }
        double mean = 0.0, var = 0.0;
        for (int64_t n = 0; n < N; n++) mean += u[n];
        mean /= (double)N;
        for (int64_t n = 0; n < N; n++) var += (u[n] - mean) * (u[n] - mean);
        var /= (double)N;
        double rstd = 1.0 / sqrt(var + (double)eps);
        for (int64_t n = 0; n < N; n++) {
            float y = (float)((u[n] - mean) * rstd);
            if (bias) y += bias[n];
            y_out[t * N + n] = y;
            if (u_out) u_out[t * N + n] = (float)u[n];
        }
    }
    free(a); free(u);
    return 0;
}

/*
 * Same forward with fp16 parameters/input, reproducing the reference's fp16
 * rounding points (every torch op on fp16 tensors rounds its result to fp16):
 *   a  = fp16(x * h)                 (:113)
 *   z  = fp16( sum_k +-a_k )          (:115)  fp16 GEMM = fp32 (here fp64) accumulate, one rounding
 *   u  = fp16(z * g)                 (:116)  in-place multiply
 *   y  = fp16((u - mean) * rstd)     (:118)  LayerNorm keeps statistics in fp32
 *   y  = fp16(y + bias)              (:119-120)
 * All buffers are uint16 bit patterns of IEEE binary16.
 */
int ob_oracle_forward_f16(const int8_t *packed, const uint16_t *x, const uint16_t *h,
                          const uint16_t *g, const uint16_t *bias, uint16_t *y_out,
                          uint16_t *u_out, int64_t T, int64_t K, int64_t N, float eps)
{
    if (K % 8 != 0) return -1;
    float *a = (float *)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    float *u = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    if (!a || !u) { free(a); free(u); return -2; }
    for (int64_t t = 0; t < T; t++) {
        for (int64_t k = 0; k < K; k++)
            a[k] = ob_round_h(ob_half_to_float(x[t * K + k]) * ob_half_to_float(h[k]));
        for (int64_t n = 0; n < N; n++) {
            const uint8_t *row = (const uint8_t *)packed + n * (K / 8);
            double acc = 0.0;
            for (int64_t k = 0; k < K; k++)
                acc += ((row[k >> 3] >> (k & 7)) & 1) ? -(double)a[k] : (double)a[k];
            float z = ob_round_h((float)acc);
            u[n] = ob_round_h(z * ob_half_to_float(g[n]));
        }
        double mean = 0.0, var = 0.0;
        for (int64_t n = 0; n < N; n++) mean += (double)u[n];
        mean /= (double)N;
        for (int64_t n = 0; n < N; n++) var += ((double)u[n] - mean) * ((double)u[n] - mean);
        var /= (double)N;
        double rstd = 1.0 / sqrt(var + (double)eps);
        for (int64_t n = 0; n < N; n++) {
            float y = ob_round_h((float)(((double)u[n] - mean) * rstd));
            if (bias) y = ob_round_h(y + ob_half_to_float(bias[n]));
            y_out[t * N + n] = ob_float_to_half(y);
            if (u_out) u_out[t * N + n] = ob_float_to_half(u[n]);
        }
    }
    free(a); free(u);
    return 0;
}

/*
 * "Reference-style" forward used only as the CPU baseline timer in bench.py:
 * like bitnet.py:98-115 it first materialises the dense +-1 matrix on every
 * call (the reference's int8_to_fp16), then runs a dense fp32 GEMV/GEMM over
 * it, then *g and LayerNorm.  scratch must hold N*K floats.  Single thread.
 */
int ob_oracle_forward_f32_unpack_every_call(const int8_t *packed, const float *x,
                                            const float *h, const float *g,
                                            float *y_out, float *scratch,
                                            int64_t T, int64_t K, int64_t N, float eps)
{
    if (K % 8 != 0) return -1;
    ob_oracle_unpack(packed, scratch, N, K);
    float *a = (float *)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    float *u = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    if (!a || !u) { free(a); free(u); return -2; }
    for (int64_t t = 0; t < T; t++) {
        for (int64_t k = 0; k < K; k++) a[k] = x[t * K + k] * h[k];
        for (int64_t n = 0; n < N; n++) {
            const float *w = scratch + n * K;
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
            int64_t k = 0;
            for (; k + 4 <= K; k += 4) {
                acc0 += a[k] * w[k]; acc1 += a[k + 1] * w[k + 1];
                acc2 += a[k + 2] * w[k + 2]; acc3 += a[k + 3] * w[k + 3];
            }
            for (; k < K; k++) acc0 += a[k] * w[k];
            u[n] = ((acc0 + acc1) + (acc2 + acc3)) * g[n];
        }
        double mean = 0.0, var = 0.0;
        for (int64_t n = 0; n < N; n++) mean += u[n];
        mean /= (double)N;
        for (int64_t n = 0; n < N; n++) var += (u[n] - mean) * (u[n] - mean);
        var /= (double)N;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        for (int64_t n = 0; n < N; n++) y_out[t * N + n] = (float)(u[n] - mean) * rstd;
    }
    free(a); free(u);
    return 0;
}

/*
 * The same reference-style forward on `threads` host cores (OpenMP): output rows are dealt to the threads, each
 * thread rebuilds ITS rows of the dense +-1 matrix (bitnet.py:98-110: still once per call, as the reference
 * does) and multiplies them -- per row the arithmetic and summation order of the single-thread function above,
 * so the results are identical.  bench.py's cpu_baseline sweeps the thread count and reports the best.
 */
int ob_oracle_forward_f32_unpack_every_call_mt(const int8_t *packed, const float *x, const float *h, const float *g,
                                               float *y_out, float *scratch, int64_t T, int64_t K, int64_t N,
                                               float eps, int threads)
{
    if (K % 8 != 0) return -1;
    if (threads < 1) threads = 1;
    float *a = (float *)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1) * (size_t)(T > 0 ? T : 1));
    float *u = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * (size_t)(T > 0 ? T : 1));
    if (!a || !u) { free(a); free(u); return -2; }
    for (int64_t t = 0; t < T; t++)
        for (int64_t k = 0; k < K; k++) a[t * K + k] = x[t * K + k] * h[k];
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t n = 0; n < N; n++) {
        float *w = scratch + n * K;
        ob_oracle_unpack(packed + n * (K / 8), w, 1, K);
        for (int64_t t = 0; t < T; t++) {
            const float *at = a + t * K;
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
            int64_t k = 0;
            for (; k + 4 <= K; k += 4) {
                acc0 += at[k] * w[k]; acc1 += at[k + 1] * w[k + 1];
                acc2 += at[k + 2] * w[k + 2]; acc3 += at[k + 3] * w[k + 3];
            }
            for (; k < K; k++) acc0 += at[k] * w[k];
            u[t * N + n] = ((acc0 + acc1) + (acc2 + acc3)) * g[n];
        }
    }
    for (int64_t t = 0; t < T; t++) {
        const float *ut = u + t * N;
        double mean = 0.0, var = 0.0;
        for (int64_t n = 0; n < N; n++) mean += ut[n];
        mean /= (double)N;
        for (int64_t n = 0; n < N; n++) var += (ut[n] - mean) * (ut[n] - mean);
        var /= (double)N;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        for (int64_t n = 0; n < N; n++) y_out[t * N + n] = (float)(ut[n] - mean) * rstd;
    }
    free(a); free(u);
    return 0;
}
