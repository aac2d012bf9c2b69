// libonebit_hip.so -- C ABI (include/onebit.h) over the gfx950 kernels.  No torch types, no
// allocation, no synchronisation: every call validates its arguments and enqueues kernels on
// the caller's stream.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/onebit.h"
#include "ob_host.h"
#include "ob_linear.h"
#include "ob_pack.h"
#include "ob_decode.h"
#include "ob_gemm.h"
#include "ob_gemm2.h"
#include "ob_gemm4.h"
#include "ob_skinny.h"
#include "ob_skinny3.h"
#include "ob_batch.h"
#include "ob_train.h"
#include "ob_flash.h"

// row kernels are instantiated per number of populated 8-half vectors per thread (ceil(width / 4096), 1..4)
#define OB_NV_DISPATCH(WIDTH, CALL1, CALL2, CALL3, CALL4) \
    do { const int nv_ = (int)(((WIDTH) + 4095) / 4096); if (nv_ <= 1) { CALL1; } else if (nv_ == 2) { CALL2; } else if (nv_ == 3) { CALL3; } else { CALL4; } } while (0)
#define OB_LAUNCH_NORM(EMBED_, WIDTH, GRID, STREAM, ARGS) \
    OB_NV_DISPATCH(WIDTH, hipLaunchKernelGGL((ob_b_norm_kernel<EMBED_, 1>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_b_norm_kernel<EMBED_, 2>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_b_norm_kernel<EMBED_, 3>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_b_norm_kernel<EMBED_, 4>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS))
#define OB_LAUNCH_SWIGLU(WIDTH, GRID, STREAM, ARGS) \
    OB_NV_DISPATCH(WIDTH, hipLaunchKernelGGL((ob_b_swiglu_kernel<1>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_b_swiglu_kernel<2>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_b_swiglu_kernel<3>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_b_swiglu_kernel<4>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS))
#define OB_LAUNCH_QKVROPE_B(BIAS_, WIDTH, GRID, STREAM, ARGS) \
    OB_NV_DISPATCH(WIDTH, hipLaunchKernelGGL((ob_qkv_rope_kernel<1, BIAS_>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_qkv_rope_kernel<2, BIAS_>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_qkv_rope_kernel<3, BIAS_>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS), \
                          hipLaunchKernelGGL((ob_qkv_rope_kernel<4, BIAS_>), GRID, dim3(OB_DEC_THREADS), 0, STREAM, ARGS))
#define OB_LAUNCH_QKVROPE(WIDTH, GRID, STREAM, ARGS) \
    do { if ((ARGS).b_q) OB_LAUNCH_QKVROPE_B(true, WIDTH, GRID, STREAM, ARGS); else OB_LAUNCH_QKVROPE_B(false, WIDTH, GRID, STREAM, ARGS); } while (0)

static thread_local char g_err[256] = "";

int ob_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ob_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ob_fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

extern "C" int onebit_abi_version(void) { return ONEBIT_ABI_VERSION; }

// Test support: fill the LDS of every CU with `pattern` (LDS keeps its contents between launches).  The decode kernels multiply
// digit images whose padding (chunks beyond K) must have been written by the launch itself; a test poisons the LDS first so that
// a launch relying on stale zeros there produces garbage instead of passing by accident (advisor finding, round 4).
__global__ __launch_bounds__(256) void ob_debug_fill_lds_kernel(uint32_t pattern, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *w = reinterpret_cast<uint32_t *>(smem);
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) w[i] = pattern;
    __syncthreads();
    if (sink && w[(threadIdx.x * 97) % (160 * 1024 / 4)] != pattern) *sink = 1;      // (keeps the stores)
}
extern "C" int onebit_debug_fill_lds(uint32_t pattern, void *stream)
{
    static bool attr_set[OB_MAX_DEVICES] = {};
    ob_set_max_lds_once(ob_debug_fill_lds_kernel, attr_set, 160 * 1024);
    hipLaunchKernelGGL(ob_debug_fill_lds_kernel, dim3(ob_cu_count() * 4), dim3(256), 160 * 1024, (hipStream_t)stream, pattern, (uint32_t *)nullptr);
    return ob_launch_status("debug_fill_lds");
}
extern "C" const char *onebit_last_error(void) { return g_err; }

// ------------------------------------------------------------------ packing --

extern "C" int onebit_pack_signs(const void *w, int dtype, void *packed, int64_t N, int64_t K,
                                 void *stream)
{
    if (N < 0 || K < 0) return ob_fail(ONEBIT_E_ARG, "pack_signs: negative size");
    if (K % 8 != 0) return ob_fail(ONEBIT_E_SHAPE, "pack_signs: K=%lld is not a multiple of 8", (long long)K);
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "pack_signs: dtype %d", dtype);
    if (N == 0 || K == 0) return 0;
    if (!w || !packed) return ob_fail(ONEBIT_E_ARG, "pack_signs: null pointer");
    const int64_t nbytes = N * (K / 8);
    const int blocks = (int)((nbytes + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ONEBIT_F16)
        hipLaunchKernelGGL(ob_pack_kernel<_Float16>, dim3(blocks), dim3(256), 0, s,
                           (const _Float16 *)w, (uint8_t *)packed, nbytes);
    else
        hipLaunchKernelGGL(ob_pack_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float *)w,
                           (uint8_t *)packed, nbytes);
    return ob_launch_status("pack_signs");
}

extern "C" int onebit_fp16_to_int8(const void *sgn, int dtype, void *packed, int64_t N, int64_t K, void *stream)
{
    if (N < 0 || K < 0) return ob_fail(ONEBIT_E_ARG, "fp16_to_int8: negative size");
    if (K % 8 != 0) return ob_fail(ONEBIT_E_SHAPE, "fp16_to_int8: K=%lld is not a multiple of 8", (long long)K);
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "fp16_to_int8: dtype %d", dtype);
    if (N == 0 || K == 0) return 0;
    if (!sgn || !packed) return ob_fail(ONEBIT_E_ARG, "fp16_to_int8: null pointer");
    if (!ob_aligned(sgn, 16)) return ob_fail(ONEBIT_E_ALIGN, "fp16_to_int8: input must be 16-byte aligned");
    const int64_t nbytes = N * (K / 8);
    const int blocks = (int)((nbytes + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ONEBIT_F16)
        hipLaunchKernelGGL(ob_f2i8_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16 *)sgn, (uint8_t *)packed, nbytes);
    else
        hipLaunchKernelGGL(ob_f2i8_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float *)sgn, (uint8_t *)packed, nbytes);
    return ob_launch_status("fp16_to_int8");
}

extern "C" int onebit_unpack_signs(const void *packed, void *out, int dtype, int64_t N, int64_t K,
                                   void *stream)
{
    if (N < 0 || K < 0) return ob_fail(ONEBIT_E_ARG, "unpack_signs: negative size");
    if (K % 8 != 0) return ob_fail(ONEBIT_E_SHAPE, "unpack_signs: K=%lld is not a multiple of 8", (long long)K);
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "unpack_signs: dtype %d", dtype);
    if (N == 0 || K == 0) return 0;
    if (!out || !packed) return ob_fail(ONEBIT_E_ARG, "unpack_signs: null pointer");
    const int64_t nbytes = N * (K / 8);
    const int blocks = (int)((nbytes + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ONEBIT_F16)
        hipLaunchKernelGGL(ob_unpack_kernel<_Float16>, dim3(blocks), dim3(256), 0, s,
                           (const uint8_t *)packed, (_Float16 *)out, nbytes);
    else
        hipLaunchKernelGGL(ob_unpack_kernel<float>, dim3(blocks), dim3(256), 0, s,
                           (const uint8_t *)packed, (float *)out, nbytes);
    return ob_launch_status("unpack_signs");
}

// ------------------------------------------------------------------ forward --

static int ob_check_linear(const char *fn, const void *packed, int64_t ldw_bytes, const void *x,
                           const void *h, int64_t T, int64_t K, int64_t N, int dtype)
{
    if (T < 0 || K < 0 || N < 0) return ob_fail(ONEBIT_E_ARG, "%s: negative size", fn);
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "%s: dtype %d", fn, dtype);
    if (K % 8 != 0) return ob_fail(ONEBIT_E_SHAPE, "%s: in_features=%lld is not a multiple of 8", fn, (long long)K);
    if (T == 0 || N == 0) return 0;
    if (K == 0) return 0;
    if (!packed || !x || !h) return ob_fail(ONEBIT_E_ARG, "%s: null pointer", fn);
    if (ldw_bytes < K / 8)
        return ob_fail(ONEBIT_E_ALIGN, "%s: packed row pitch %lld < K/8", fn, (long long)ldw_bytes);
    if (!ob_aligned(x, 16) || !ob_aligned(h, 16))
        return ob_fail(ONEBIT_E_ALIGN, "%s: x and h must be 16-byte aligned", fn);
    if (T > 0x7fffffffLL / 16 || N > 0x7fffffffLL / 16 || K > 0x7fffffffLL)
        return ob_fail(ONEBIT_E_ARG, "%s: dimension too large", fn);
    return 0;
}

// The MFMA path needs whole, dword-aligned packed words and 16-byte aligned activation rows.
static inline bool ob_mfma_ok(const void *packed, int64_t ldw_bytes, int64_t ldx, int64_t K, int dtype)
{
    return dtype == ONEBIT_F16 && K % 32 == 0 && ldw_bytes % 4 == 0 && ob_aligned(packed, 4) && ldx % 8 == 0;
}

template <typename TX>
static void ob_launch_simple(const void *packed, int64_t ldw_bytes, const void *x, int64_t ldx,
                             const void *h, float *zp, int64_t T, int64_t K, int64_t N, hipStream_t s)
{
    hipLaunchKernelGGL(ob_simple_kernel<TX>, dim3((unsigned)N, (unsigned)T), dim3(64), 0, s,
                       (const uint8_t *)packed, ldw_bytes, (const TX *)x, ldx, (const TX *)h, zp,
                       (int)T, (int)K, (int)N);
}

static inline size_t ob_skinny_lds(int rt) { return OB_SKINNY_LDS(rt); }

#ifdef OB_PROFILE_STAMPS
static unsigned long long *g_dbg = nullptr;
static unsigned long long *ob_dbg_buffer()
{
    if (!getenv("OB_TIMING")) return nullptr;
    if (!g_dbg) { (void)hipMalloc(&g_dbg, 4096 * 128 * sizeof(unsigned long long)); (void)hipMemset(g_dbg, 0, 4096 * 128 * sizeof(unsigned long long)); }
    return g_dbg;
}
#endif

template <bool PARTIAL, int RT, int RNT = 4>
static void ob_launch_skinny(const ObSkinnyArgs &ka_in, int tiles, hipStream_t s)
{
    ObSkinnyArgs ka = ka_in;
#ifdef OB_PROFILE_STAMPS
    ka.dbg = ob_dbg_buffer();
#endif
    const size_t lds = OB_SKINNY_LDS2(RT, RNT) > (size_t)RT * 32768 ? OB_SKINNY_LDS2(RT, RNT) : (size_t)RT * 32768;
    static bool attr_set[OB_MAX_DEVICES] = {};
    ob_set_max_lds_once(ob_skinny_f16_kernel<PARTIAL, RT, RNT>, attr_set, (int)lds);
    hipLaunchKernelGGL((ob_skinny_f16_kernel<PARTIAL, RT, RNT>), dim3((unsigned)tiles), dim3(512), lds, s, ka);
}

static inline bool ob_skinny_ok(const void *packed, int64_t ldw_bytes, int64_t T, int64_t K)
{
    static const int skinny_env = getenv("OB_SKINNY") ? atoi(getenv("OB_SKINNY")) : 1;
    return skinny_env && T >= 2 && T <= 64 && K % 128 == 0 && K >= 512 && ldw_bytes % 16 == 0 && ob_aligned(packed, 16);
}

// z (fp32 partial, PARTIAL) or u (fp16) for T tokens.
template <bool PARTIAL>
static void ob_launch_mm16(const void *packed, int64_t ldw_bytes, const void *x, int64_t ldx,
                           const void *h, const void *g, void *u, float *zp, int64_t T, int64_t K,
                           int64_t N, hipStream_t s)
{
    if (T >= 2) {       // MFMA GEMM: token tile sized to T (16 / 32 / 64 tokens x 64 rows, or 128 x 128 for prefill)
#define OB_GEMM_GO(WN_, WT_, RN_, RT_)                                                                         \
        do {                                                                                                   \
            const int nbn = (int)((N + WN_ * RN_ * 16 - 1) / (WN_ * RN_ * 16));                                \
            const int nbt = (int)((T + WT_ * RT_ * 16 - 1) / (WT_ * RT_ * 16));                                \
            hipLaunchKernelGGL((ob_gemm_f16_kernel<PARTIAL, WN_, WT_, RN_, RT_>), dim3((unsigned)(nbn * nbt)), \
                               dim3(WN_ * WT_ * 64), 0, s, (const uint32_t *)packed, ldw_bytes / 4,            \
                               (const _Float16 *)x, ldx, (const _Float16 *)h, (const _Float16 *)g,             \
                               (_Float16 *)u, zp, (int)T, (int)K, (int)N, nbn, nbt);                           \
        } while (0)
        // 2 <= T <= 64 with 16-byte aligned packed rows and K % 128 == 0: the phase-prefetched skinny kernel
        const bool skinny = ob_skinny_ok(packed, ldw_bytes, T, K);
#define OB_SKINNY_GO(RT_)                                                                                      \
        do {                                                                                                   \
            ObSkinnyArgs ka = {};                                                                              \
            ka.p[0] = {(const uint32_t *)packed, (long long)(ldw_bytes / 4), (const _Float16 *)h,              \
                       (const _Float16 *)g, (const _Float16 *)x, (_Float16 *)u, zp, (int)N, (int)K,            \
                       (int)((N + 63) / 64)};                                                                  \
            ka.p[1] = ka.p[0]; ka.p[2] = ka.p[0];                                                              \
            ka.ldx = ldx; ka.T = (int)T;                                                                       \
            ob_launch_skinny<PARTIAL, RT_>(ka, ka.p[0].tile_end, s);                                           \
        } while (0)
        if (skinny && T <= 16) OB_SKINNY_GO(1);
        else if (skinny && T <= 32) OB_SKINNY_GO(2);
        else if (skinny) OB_SKINNY_GO(4);
        else if (T <= 16) OB_GEMM_GO(4, 1, 1, 1);
        else if (T <= 32) OB_GEMM_GO(4, 1, 1, 2);
        else if (T <= 64) OB_GEMM_GO(4, 1, 1, 4);
        else {
            // large T without a workspace: the 128 x 128 kernel.  The register-staged 256 x 256 / 8-wave kernel (ob_gemm2.h) issues
            // its staging loads from inline asm whose in-flight destination registers the compiler cannot model, so it is opt-in
            // (OB_GEMM2=1: from 4 rounds of tiles, =2: forced on any eligible shape -- the forced-route parity tests); every product
            // caller passes a workspace and takes the LDS-DMA kernel instead
            static const int gemm2_env = getenv("OB_GEMM2") ? atoi(getenv("OB_GEMM2")) : 0;
            // (worth it from ~4 rounds of 256 x 256 tiles over the CUs; smaller problems quantise badly)
            const int64_t tiles2 = ((N + OB_G2_N - 1) / OB_G2_N) * ((T + OB_G2_T - 1) / OB_G2_T);
            if (gemm2_env && (tiles2 >= 4 * (int64_t)ob_cu_count() || gemm2_env == 2) && T >= 192 && K % OB_G2_K == 0 && N % 4 == 0 && ldx % 8 == 0) {
                static bool attr_set[OB_MAX_DEVICES] = {};
                ob_set_max_lds_once(ob_gemm2_f16_kernel<PARTIAL>, attr_set, OB_G2_LDS);
                const int nbn = (int)((N + OB_G2_N - 1) / OB_G2_N), nbt = (int)((T + OB_G2_T - 1) / OB_G2_T);
                hipLaunchKernelGGL((ob_gemm2_f16_kernel<PARTIAL>), dim3((unsigned)(nbn * nbt)), dim3(OB_G2_THREADS), OB_G2_LDS, s,
                                   (const uint32_t *)packed, ldw_bytes / 4, (const _Float16 *)x, ldx, (const _Float16 *)h,
                                   (const _Float16 *)g, (_Float16 *)u, zp, (int)T, (int)K, (int)N, nbn);
            } else OB_GEMM_GO(2, 2, 4, 4);
        }
#undef OB_SKINNY_GO
#undef OB_GEMM_GO
        return;
    }
    const int fast = (ldw_bytes % 16 == 0) && ob_aligned(packed, 16) && (ldx % 8 == 0);
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)((T + 15) / 16));
    const int64_t units = fast ? (K + 511) / 512 : (K + 127) / 128;
#define OB_MM16(WV)                                                                              \
    hipLaunchKernelGGL((ob_mm16_f16_kernel<WV, PARTIAL>), grid, dim3(WV * 64), 0, s,             \
                       (const uint32_t *)packed, ldw_bytes / 4, (const _Float16 *)x, ldx,         \
                       (const _Float16 *)h, (const _Float16 *)g, (_Float16 *)u, zp, (int)T,       \
                       (int)K, (int)N, fast)
    if (units >= 8) OB_MM16(8);
    else if (units >= 4) OB_MM16(4);
    else if (units >= 2) OB_MM16(2);
    else OB_MM16(1);
#undef OB_MM16
}

// fp16 LayerNorm epilogue: rows-in-registers kernel when the shape allows, else the generic one
template <bool FROM_Z>
static void ob_launch_ln_f16(const float *z, const _Float16 *uin, const _Float16 *g, const _Float16 *bias,
                             _Float16 *y, _Float16 *uout, int64_t T, int64_t N, float eps, int skip, hipStream_t s)
{
    const bool vec = N % 8 == 0 && N <= 16384 && ob_aligned(y, 16) && (!uin || ob_aligned(uin, 16)) &&
                     (!uout || ob_aligned(uout, 16)) && (!z || ob_aligned(z, 16)) && (!g || ob_aligned(g, 16)) &&
                     (!bias || ob_aligned(bias, 16));
    if (vec)
        hipLaunchKernelGGL((ob_layernorm_rows_kernel<FROM_Z>), dim3((unsigned)T), dim3(256), 0, s, z, uin, g, bias, y, uout, (int)N, eps, skip);
    else
        hipLaunchKernelGGL((ob_layernorm_kernel<_Float16, FROM_Z>), dim3((unsigned)T), dim3(256), 0, s, z, uin, g, bias, y, uout, (int)N, eps, skip);
}

struct ObGemvArgs;
struct ObSk3Args;
template <bool PARTIAL> static bool ob_launch_skinny3(const ObSk3Args &a, int grid, int rnt, hipStream_t s);
static bool ob_skinny3_shape_ok(const void *packed, int64_t ldw_bytes, const void *a, int64_t lda, int64_t T, int64_t K, int64_t N);
static int ob_skinny3_rnt(int64_t N);
static int ob_skinny3_pick_rnt(const int64_t *N, int np, int copies);
static int ob_launch_dec_gemv(const ObGemvArgs &a_in, hipStream_t s);
static int ob_single_token_gemv(const void *packed, int64_t ldw_bytes, const void *x, const void *h, const void *g,
                                void *u, int64_t K, int64_t N, hipStream_t s);

// The LDS-DMA prefill GEMM (ob_gemm3_f16_kernel) consumes pre-scaled activations from the caller's
// workspace (or the producer's rows).  OB_GEMM3=0 disables it (A/B), =2 forces it on any eligible shape.
// Eligibility: since the 4-wave form (256 rows x 128 tokens, two workgroups per CU) it wins from a grid of about two
// thirds of the CU count -- measured with the scaling pass included, default dispatch vs forced (tools/gemm_route_probe.py):
// 172 workgroups ([512, 4096] -> 11008) 664 vs 569 TFLOP/s, 256 ([2048, 4096] -> 4096) 863 vs 787, 344 ([1024, 4096] ->
// 11008) 856 vs 693, 516 ([1536, 4096] -> 11008) 863 vs 683; 128 workgroups ([1024, 4096] -> 4096) 494 vs 609: not taken.
static bool ob_gemm3_ok(int64_t T, int64_t K, int64_t N)
{
    static const int env = getenv("OB_GEMM3") ? atoi(getenv("OB_GEMM3")) : 1;
    // whole quads of K steps; 32-bit byte offsets from the base pointers inside the kernel
    if (!env || T < 192 || K % (4 * OB_G2_K) != 0 || N % 4 != 0 || T * K * 2 >= ((int64_t)1 << 32) || N * (K / 8) >= ((int64_t)1 << 32)) return false;
    const int64_t tiles = ((N + OB_G2_N - 1) / OB_G2_N) * ((T + 127) / 128);
    return 3 * tiles >= 2 * (int64_t)ob_cu_count() || env == 2;
}

// The LDS-DMA GEMM in its 8-wave (256 x 256 tile, one workgroup per CU) or 4-wave (256 rows x 128 tokens, two workgroups per
// CU) form; OB_GEMM3_WT=1 / 2 selects (A/B), default below.
#ifndef OB_GEMM4_DEFAULT
#define OB_GEMM4_DEFAULT 0
#endif
template <bool PARTIAL>
static void ob_launch_gemm3(const uint32_t *W, int64_t ldw_words, const _Float16 *a, int64_t lda, const _Float16 *g, _Float16 *u,
                            float *zp, int64_t T, int64_t K, int64_t N, hipStream_t s)
{
    static const int wt_env = getenv("OB_GEMM3_WT") ? atoi(getenv("OB_GEMM3_WT")) : 1;    // 4-wave form: +3-4 % (1286 -> 1335 TFLOP/s on 4096 -> 11008), bit-identical
    static const int g4_env = getenv("OB_GEMM4") ? atoi(getenv("OB_GEMM4")) : OB_GEMM4_DEFAULT;   // the 32x32x16-MFMA form of the same tiling (ob_gemm4.h)
    const int nbn = (int)((N + OB_G2_N - 1) / OB_G2_N);
    if (g4_env) {
        static bool attr_set[OB_MAX_DEVICES] = {};
        ob_set_max_lds_once(ob_gemm4_f16_kernel<PARTIAL>, attr_set, OB_G4_LDS);
        const int nbt = (int)((T + 127) / 128);
        hipLaunchKernelGGL((ob_gemm4_f16_kernel<PARTIAL>), dim3((unsigned)(nbn * nbt)), dim3(256), OB_G4_LDS, s,
                           W, ldw_words, a, lda, g, u, zp, (int)T, (int)K, (int)N, nbn);
    } else if (wt_env == 1) {
        static bool attr_set[OB_MAX_DEVICES] = {};
        ob_set_max_lds_once(ob_gemm3_f16_kernel<PARTIAL, 1>, attr_set, OB_G3_LDS_W(1));
        const int nbt = (int)((T + 127) / 128);
        hipLaunchKernelGGL((ob_gemm3_f16_kernel<PARTIAL, 1>), dim3((unsigned)(nbn * nbt)), dim3(256), OB_G3_LDS_W(1), s,
                           W, ldw_words, a, lda, g, u, zp, (int)T, (int)K, (int)N, nbn);
    } else {
        static bool attr_set[OB_MAX_DEVICES] = {};
        ob_set_max_lds_once(ob_gemm3_f16_kernel<PARTIAL, 2>, attr_set, OB_G3_LDS);
        const int nbt = (int)((T + OB_G2_T - 1) / OB_G2_T);
        hipLaunchKernelGGL((ob_gemm3_f16_kernel<PARTIAL, 2>), dim3((unsigned)(nbn * nbt)), dim3(OB_G2_THREADS), OB_G3_LDS, s,
                           W, ldw_words, a, lda, g, u, zp, (int)T, (int)K, (int)N, nbn);
    }
}

// bytes a call cannot do without (F32: fp32 z is staged in y itself; F16 on the MFMA path: u is staged in
// y; F16 shapes the MFMA path cannot take (K % 32 != 0) stage fp32 z in the workspace)
template <bool ZIN>
static void ob_launch_ln_rows(const ObLnRowsArgs &la, int64_t T, hipStream_t s)
{
    const int nv = (la.N + OB_DEC_THREADS * 8 - 1) / (OB_DEC_THREADS * 8);
#define OB_LN_ROWS(NV_) do { if (la.bias) hipLaunchKernelGGL((ob_ln_rows_kernel<NV_, true, ZIN>), dim3((unsigned)T), dim3(OB_DEC_THREADS), 0, s, la); \
                             else hipLaunchKernelGGL((ob_ln_rows_kernel<NV_, false, ZIN>), dim3((unsigned)T), dim3(OB_DEC_THREADS), 0, s, la); } while (0)
    if (nv == 1) OB_LN_ROWS(1); else if (nv == 2) OB_LN_ROWS(2); else if (nv == 3) OB_LN_ROWS(3); else OB_LN_ROWS(4);
#undef OB_LN_ROWS
}

static inline size_t ob_align256(size_t b) { return (b + 255) & ~(size_t)255; }
// shape part of ob_gemm3_ksplit_n's test (pointers unknown: onebit_linear_workspace_bytes)
static bool ob_ksplit_shape_ok(int64_t T, int64_t K, int64_t N)
{
    onebit_proj_t p = {};
    static const char dummy[16] __attribute__((aligned(16))) = {};
    p.weight = dummy; p.weight_scale = dummy; p.input_factor = dummy; p.N = N; p.K = K; p.ldw_bytes = K / 8;
    return K % 128 == 0 && N % 8 == 0 && N <= OB_DEC_MAXV * OB_DEC_THREADS * 8 && ob_gemm3_ksplit_n(p, T) > 0;
}

static size_t ob_required_workspace(int64_t T, int64_t K, int64_t N, int dtype)
{
    if (T <= 0 || N <= 0) return 0;
    if (dtype == ONEBIT_F16 && K % 32 != 0) return (size_t)T * (size_t)N * sizeof(float);
    return 0;
}

extern "C" size_t onebit_linear_workspace_bytes(int64_t T, int64_t K, int64_t N, int dtype)
{
    if (T <= 0 || N <= 0) return 0;
    // large prefill calls run fastest with room for the pre-scaled activations a = fp16(x * h) (LDS-DMA
    // GEMM); a smaller or absent workspace is accepted and selects the register-staged kernel
    if (dtype == ONEBIT_F16 && K % 32 == 0 && ob_gemm3_ok(T, K, N)) return (size_t)T * (size_t)K * 2;
    // a few hundred rows of a projection whose tiles alone leave the chip idle: room for the pre-scaled rows AND the fp32 sums of up to
    // four K-slices (ob_gemm3_ksplit); without it such a call keeps the register-staged kernel
    if (dtype == ONEBIT_F16 && ob_ksplit_shape_ok(T, K, N)) return ob_align256((size_t)T * (size_t)K * 2) + 4 * ob_align256((size_t)T * (size_t)N * 4);
    return ob_required_workspace(T, K, N, dtype);
}

extern "C" int onebit_linear_forward(const void *packed, int64_t ldw_bytes, const void *x,
                                     const void *h, const void *g, const void *bias, void *y,
                                     void *u_or_null, void *workspace, size_t workspace_bytes,
                                     int64_t T, int64_t K, int64_t N, int dtype, float ln_eps,
                                     unsigned flags, void *stream)
{
    int rc = ob_check_linear("linear_forward", packed, ldw_bytes, x, h, T, K, N, dtype);
    if (rc) return rc;
    if (flags & ~(ONEBIT_FLAG_SKIP_LN | ONEBIT_FLAG_PRESCALED | ONEBIT_FLAG_TILE_STATS)) return ob_fail(ONEBIT_E_FLAG, "linear_forward: unknown flags 0x%x", flags);
    const bool prescaled = (flags & ONEBIT_FLAG_PRESCALED) != 0;
    if (prescaled && !(onebit_linear_prescaled_ok(T, K, N, dtype) && ldw_bytes % 16 == 0 && ob_aligned(packed, 16) && ob_aligned(x, 16) &&
                       N * ldw_bytes < ((int64_t)1 << 32)))
        return ob_fail(ONEBIT_E_FLAG, "linear_forward: ONEBIT_FLAG_PRESCALED on a call that does not take the LDS-DMA GEMM "
                                      "(ask onebit_linear_prescaled_ok first)");
    if (workspace_bytes < ob_required_workspace(T, K, N, dtype))
        return ob_fail(ONEBIT_E_WSPACE, "linear_forward: workspace too small");
    if (T == 0 || N == 0) return 0;
    if (!g || !y) return ob_fail(ONEBIT_E_ARG, "linear_forward: null pointer");
    if (!ob_aligned(g, 2) || !ob_aligned(y, 16)) return ob_fail(ONEBIT_E_ALIGN, "linear_forward: y must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int skip = (flags & ONEBIT_FLAG_SKIP_LN) ? 1 : 0;
    // ONEBIT_FLAG_TILE_STATS (with SKIP_LN, fp16): u_or_null is NOT a second output but fp32 [T, N / 64, 2] and receives the
    // LayerNorm partials of the rows out of the LDS-DMA GEMM's epilogue; the call must take that GEMM (onebit_linear_tile_stats_ok)
    float *tile_stats = nullptr;
    bool tile_stats_done = false;
    if (flags & ONEBIT_FLAG_TILE_STATS) {
        if (!skip || dtype != ONEBIT_F16 || !u_or_null || !ob_aligned(u_or_null, 8) || !onebit_linear_tile_stats_ok(T, K, N, dtype))
            return ob_fail(ONEBIT_E_FLAG, "linear_forward: ONEBIT_FLAG_TILE_STATS needs SKIP_LN, fp16, a statistics buffer and a shape "
                                          "onebit_linear_tile_stats_ok accepts");
        tile_stats = (float *)u_or_null;
        u_or_null = nullptr;
    }
    if (dtype == ONEBIT_F16 && (K == 0 || ob_mfma_ok(packed, ldw_bytes, K, K, dtype))) {
        _Float16 *ubuf = (_Float16 *)(u_or_null ? u_or_null : y);
        if (K == 0) {
            (void)hipMemsetAsync(ubuf, 0, (size_t)T * N * 2, s);
        } else if (T == 1 && K % 128 == 0 && ldw_bytes % 16 == 0 && ob_aligned(packed, 16) && K <= 16384) {
            // one token: the persistent decode GEMV (plain prologue) instead of the 16-token tile kernel
            rc = ob_single_token_gemv(packed, ldw_bytes, x, h, g, ubuf, K, N, s);
            if (rc) return rc;
        } else if (prescaled && !ob_gemm3_ok(T, K, N)) {
            // 2 <= T <= 64 on producer-scaled rows: the LDS-DMA skinny GEMM (ob_skinny3.h)
            if (!ob_skinny3_shape_ok(packed, ldw_bytes, x, K, T, K, N))
                return ob_fail(ONEBIT_E_FLAG, "linear_forward: ONEBIT_FLAG_PRESCALED on a call no pre-scaled kernel takes");
            const int rnt = ob_skinny3_rnt(N);
            ObSk3Args a = {};
            a.lda = K; a.T = (int)T;
            a.p[0] = {(const uint32_t *)packed, (long long)(ldw_bytes / 4), (const _Float16 *)g, (const _Float16 *)x, ubuf, nullptr, nullptr,
                      (int)N, (int)K, (int)((N + 16 * rnt - 1) / (16 * rnt))};
            a.p[1] = a.p[0]; a.p[2] = a.p[0];
            if (!ob_launch_skinny3<false>(a, a.p[0].wg_end, rnt, s))
                return ob_fail(ONEBIT_E_FLAG, "linear_forward: ONEBIT_FLAG_PRESCALED on a call no pre-scaled kernel takes");
            rc = ob_launch_status("linear_forward(skinny3)");
            if (rc) return rc;
        } else if (prescaled || (ob_gemm3_ok(T, K, N) && workspace && workspace_bytes >= (size_t)T * (size_t)K * 2 && ob_aligned(workspace, 16) &&
                                 ldw_bytes % 16 == 0 && ob_aligned(packed, 16) && N * ldw_bytes < ((int64_t)1 << 32))) {
            // pre-scale once (the fp16 rounding of bitnet.py:113) -- unless the producer of x already did
            // (ONEBIT_FLAG_PRESCALED) --, then the LDS-DMA GEMM on the scaled rows
            const _Float16 *a = (const _Float16 *)x;
            if (!prescaled) {
                const int64_t nvec = T * K / 8;
                hipLaunchKernelGGL(ob_scale_rows_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const _Float16 *)x, K,
                                   (const _Float16 *)h, (_Float16 *)workspace, T, (int)K);
                a = (const _Float16 *)workspace;
            }
            ob_launch_gemm3<false>((const uint32_t *)packed, ldw_bytes / 4, a, K, (const _Float16 *)g, ubuf, tile_stats, T, K, N, s);
            tile_stats_done = tile_stats != nullptr;
            rc = ob_launch_status("linear_forward(gemm3)");
            if (rc) return rc;
        } else {
            // K-sliced LDS-DMA GEMM (ob_gemm3_ksplit) when the caller's workspace has room for the pre-scaled rows and the slices' fp32 sums:
            // scale pass, ns short rounds of workgroups in one launch, then ONE row pass that adds the slices, applies fp16(fp16(.) * g)
            // and the LayerNorm -- whole call, 512 rows: 11008 -> 4096 132 -> 65 us, 4096 -> 4096 57 -> 44 us (tools/module_T_sweep.py)
            onebit_proj_t pj = {};
            pj.weight = packed; pj.input_factor = h; pj.weight_scale = g; pj.N = N; pj.K = K; pj.ldw_bytes = ldw_bytes;
            const int ns = (!tile_stats && N % 8 == 0 && N <= OB_DEC_MAXV * OB_DEC_THREADS * 8 && (!bias || ob_aligned(bias, 16))) ? ob_gemm3_ksplit_n(pj, T) : 0;
            const size_t a_bytes = ob_align256((size_t)T * (size_t)K * 2), z_bytes = ob_align256((size_t)T * (size_t)N * 4);
            if (ns > 0 && workspace && ob_aligned(workspace, 256) && workspace_bytes >= a_bytes + (size_t)ns * z_bytes) {
                const int64_t nvec = T * K / 8;
                hipLaunchKernelGGL(ob_scale_rows_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const _Float16 *)x, K,
                                   (const _Float16 *)h, (_Float16 *)workspace, T, (int)K);
                float *zs[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int j = 0; j < ns; ++j) zs[j] = (float *)((char *)workspace + a_bytes + (size_t)j * z_bytes);
                if ((rc = ob_gemm3_ksplit(pj, workspace, zs, ns, T, s))) return rc;
                ObLnRowsArgs la = {};
                for (int j = 0; j < ns; ++j) la.z[j] = zs[j];
                la.g = (const _Float16 *)g; la.bias = skip ? nullptr : (const _Float16 *)bias; la.N = (int)N; la.eps = ln_eps;
                if (u_or_null && !skip) {                       // both outputs wanted: u first, then y from it
                    la.y = (_Float16 *)u_or_null; la.skip = 1;
                    ob_launch_ln_rows<true>(la, T, s);
                    ObLnRowsArgs lb = {};
                    lb.uin = (const _Float16 *)u_or_null; lb.bias = (const _Float16 *)bias; lb.y = (_Float16 *)y; lb.N = (int)N; lb.eps = ln_eps;
                    ob_launch_ln_rows<false>(lb, T, s);
                } else {
                    la.y = (_Float16 *)(skip ? ubuf : y); la.skip = skip;
                    ob_launch_ln_rows<true>(la, T, s);
                    if (skip && ubuf != y) (void)hipMemcpyAsync(y, ubuf, (size_t)T * N * 2, hipMemcpyDeviceToDevice, s);
                }
                return ob_launch_status("linear_forward(k-sliced)");
            }
            ob_launch_mm16<false>(packed, ldw_bytes, x, K, h, g, ubuf, nullptr, T, K, N, s);
            rc = ob_launch_status("linear_forward(mm16)");
            if (rc) return rc;
        }
        if (tile_stats && !tile_stats_done)
            return ob_fail(ONEBIT_E_FLAG, "linear_forward: ONEBIT_FLAG_TILE_STATS on a call that did not take the LDS-DMA GEMM "
                                          "(workspace for the pre-scaled rows missing?)");
        if (skip && ubuf == y) return 0;
        static const int ln_v3 = getenv("OB_LN_ROWS") ? atoi(getenv("OB_LN_ROWS")) : 1;           // A/B: 0 = ob_layernorm_rows_kernel
        if (ln_v3 && !skip && T >= 64 && N % 8 == 0 && N <= OB_DEC_MAXV * OB_DEC_THREADS * 8 && ob_aligned(ubuf, 16) && (!bias || ob_aligned(bias, 16))) {
            ObLnRowsArgs la = {};
            la.uin = (const _Float16 *)ubuf; la.bias = (const _Float16 *)bias; la.y = (_Float16 *)y; la.N = (int)N; la.eps = ln_eps;
            ob_launch_ln_rows<false>(la, T, s);
            return ob_launch_status("linear_forward(layernorm rows)");
        }
        ob_launch_ln_f16<false>(nullptr, (const _Float16 *)ubuf, (const _Float16 *)g, (const _Float16 *)bias,
                                (_Float16 *)y, nullptr, T, N, ln_eps, skip, s);
        return ob_launch_status("linear_forward(layernorm)");
    }
    if (dtype == ONEBIT_F16) {
        // generic shapes: fp32 z in the caller's workspace, then the shared epilogue
        const size_t need = (size_t)T * N * sizeof(float);
        if (!workspace || workspace_bytes < need || !ob_aligned(workspace, 16))
            return ob_fail(ONEBIT_E_WSPACE, "linear_forward: needs %zu bytes of 16-byte aligned workspace", need);
        ob_launch_simple<_Float16>(packed, ldw_bytes, x, K, h, (float *)workspace, T, K, N, s);
        rc = ob_launch_status("linear_forward(simple)");
        if (rc) return rc;
        ob_launch_ln_f16<true>((const float *)workspace, nullptr, (const _Float16 *)g, (const _Float16 *)bias,
                               (_Float16 *)y, (_Float16 *)u_or_null, T, N, ln_eps, skip, s);
        return ob_launch_status("linear_forward(layernorm)");
    }
    // F32: z (fp32) staged in y, then g / LayerNorm in place.
    float *zbuf = (float *)y;
    if (K == 0) {
        (void)hipMemsetAsync(zbuf, 0, (size_t)T * N * 4, s);
    } else {
        ob_launch_simple<float>(packed, ldw_bytes, x, K, h, zbuf, T, K, N, s);
        rc = ob_launch_status("linear_forward(simple)");
        if (rc) return rc;
    }
    hipLaunchKernelGGL((ob_layernorm_kernel<float, true>), dim3((unsigned)T), dim3(256), 0, s,
                       (const float *)zbuf, (const float *)nullptr, (const float *)g,
                       (const float *)bias, (float *)y, (float *)u_or_null, (int)N, ln_eps, skip);
    return ob_launch_status("linear_forward(layernorm)");
}

extern "C" int onebit_matmul_partial_ws(const void *packed, int64_t ldw_bytes, const void *x,
                                        int64_t ldx, const void *h, float *zp, void *workspace, size_t workspace_bytes,
                                        int64_t T, int64_t K, int64_t N, int dtype, void *stream)
{
    int rc = ob_check_linear("matmul_partial", packed, ldw_bytes, x, h, T, K, N, dtype);
    if (rc) return rc;
    if (T == 0 || N == 0) return 0;
    if (!zp) return ob_fail(ONEBIT_E_ARG, "matmul_partial: null pointer");
    if (ldx < K) return ob_fail(ONEBIT_E_ARG, "matmul_partial: ldx < K");
    hipStream_t s = (hipStream_t)stream;
    if (K == 0) {
        (void)hipMemsetAsync(zp, 0, (size_t)T * N * 4, s);
        return 0;
    }
    if (ob_mfma_ok(packed, ldw_bytes, ldx, K, dtype) && ob_gemm3_ok(T, K, N) && workspace && workspace_bytes >= (size_t)T * (size_t)K * 2 &&
        ob_aligned(workspace, 16) && ldw_bytes % 16 == 0 && ob_aligned(packed, 16) && N * ldw_bytes < ((int64_t)1 << 32)) {
        // LDS-DMA GEMM on the pre-scaled K slice (see onebit_linear_forward)
        _Float16 *a = (_Float16 *)workspace;
        const int64_t nvec = T * K / 8;
        hipLaunchKernelGGL(ob_scale_rows_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, (const _Float16 *)x, ldx,
                           (const _Float16 *)h, a, T, (int)K);
        ob_launch_gemm3<true>((const uint32_t *)packed, ldw_bytes / 4, (const _Float16 *)a, K, nullptr, nullptr, zp, T, K, N, s);
    } else if (ob_mfma_ok(packed, ldw_bytes, ldx, K, dtype))
        ob_launch_mm16<true>(packed, ldw_bytes, x, ldx, h, nullptr, nullptr, zp, T, K, N, s);
    else if (dtype == ONEBIT_F16)
        ob_launch_simple<_Float16>(packed, ldw_bytes, x, ldx, h, zp, T, K, N, s);
    else
        ob_launch_simple<float>(packed, ldw_bytes, x, ldx, h, zp, T, K, N, s);
    return ob_launch_status("matmul_partial");
}

extern "C" int onebit_matmul_partial(const void *packed, int64_t ldw_bytes, const void *x,
                                     int64_t ldx, const void *h, float *zp, int64_t T, int64_t K,
                                     int64_t N, int dtype, void *stream)
{
    return onebit_matmul_partial_ws(packed, ldw_bytes, x, ldx, h, zp, nullptr, 0, T, K, N, dtype, stream);
}

extern "C" int onebit_scale_layernorm(const float *z, const void *g, const void *bias, void *y,
                                      void *u_or_null, int64_t T, int64_t N, int dtype,
                                      float ln_eps, unsigned flags, void *stream)
{
    if (T < 0 || N < 0) return ob_fail(ONEBIT_E_ARG, "scale_layernorm: negative size");
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "scale_layernorm: dtype %d", dtype);
    if (flags & ~ONEBIT_FLAG_SKIP_LN) return ob_fail(ONEBIT_E_FLAG, "scale_layernorm: unknown flags 0x%x", flags);
    if (T == 0 || N == 0) return 0;
    if (!z || !g || !y) return ob_fail(ONEBIT_E_ARG, "scale_layernorm: null pointer");
    if (T > 0x7fffffffLL || N > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "scale_layernorm: dimension too large");
    hipStream_t s = (hipStream_t)stream;
    const int skip = (flags & ONEBIT_FLAG_SKIP_LN) ? 1 : 0;
    if (dtype == ONEBIT_F16)
        ob_launch_ln_f16<true>(z, nullptr, (const _Float16 *)g, (const _Float16 *)bias, (_Float16 *)y,
                               (_Float16 *)u_or_null, T, N, ln_eps, skip, s);
    else
        hipLaunchKernelGGL((ob_layernorm_kernel<float, true>), dim3((unsigned)T), dim3(256), 0, s, z,
                           (const float *)nullptr, (const float *)g, (const float *)bias, (float *)y,
                           (float *)u_or_null, (int)N, ln_eps, skip);
    return ob_launch_status("scale_layernorm");
}

extern "C" int onebit_linear_tile_stats_ok(int64_t T, int64_t K, int64_t N, int dtype)
{
    return (dtype == ONEBIT_F16 && T > 0 && K > 0 && N > 0 && N % 64 == 0 && K % 32 == 0 && ob_gemm3_ok(T, K, N)) ? 1 : 0;
}

extern "C" int onebit_tile_stats_combine(const float *tiles, float *stats, int64_t T, int64_t N, void *stream)
{
    if (T < 0 || N <= 0 || N % 64 != 0) return ob_fail(ONEBIT_E_SHAPE, "tile_stats_combine: N must be a positive multiple of 64");
    if (T == 0) return 0;
    if (!tiles || !stats || !ob_aligned(tiles, 8)) return ob_fail(ONEBIT_E_ARG, "tile_stats_combine: null or misaligned pointer");
    hipLaunchKernelGGL(ob_tile_stats_combine_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, tiles, stats,
                       (int)T, (int)(N / 64));
    return ob_launch_status("tile_stats_combine");
}

extern "C" int onebit_row_stats(const void *u, float *stats, int64_t T, int64_t n, int dtype, void *stream)
{
    if (T < 0 || n < 0) return ob_fail(ONEBIT_E_ARG, "row_stats: negative size");
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "row_stats: dtype %d", dtype);
    if (T == 0) return 0;
    if (n == 0) return ob_fail(ONEBIT_E_SHAPE, "row_stats: empty rows");
    if (!u || !stats) return ob_fail(ONEBIT_E_ARG, "row_stats: null pointer");
    if (T > 0x7fffffffLL || n > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "row_stats: dimension too large");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ONEBIT_F16)
        hipLaunchKernelGGL((ob_row_stats_kernel<_Float16>), dim3((unsigned)T), dim3(256), 0, s, (const _Float16 *)u, stats, (int)n);
    else
        hipLaunchKernelGGL((ob_row_stats_kernel<float>), dim3((unsigned)T), dim3(256), 0, s, (const float *)u, stats, (int)n);
    return ob_launch_status("row_stats");
}

extern "C" int onebit_normalize_rows(const void *u, const float *mean, const float *rstd, const void *bias,
                                     void *y, int64_t T, int64_t n, int dtype, void *stream)
{
    if (T < 0 || n < 0) return ob_fail(ONEBIT_E_ARG, "normalize_rows: negative size");
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "normalize_rows: dtype %d", dtype);
    if (T == 0 || n == 0) return 0;
    if (!u || !mean || !rstd || !y) return ob_fail(ONEBIT_E_ARG, "normalize_rows: null pointer");
    if (T > 0x7fffffffLL || n > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "normalize_rows: dimension too large");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ONEBIT_F16)
        hipLaunchKernelGGL((ob_normalize_rows_kernel<_Float16>), dim3((unsigned)T), dim3(256), 0, s, (const _Float16 *)u,
                           mean, rstd, (const _Float16 *)bias, (_Float16 *)y, (int)n);
    else
        hipLaunchKernelGGL((ob_normalize_rows_kernel<float>), dim3((unsigned)T), dim3(256), 0, s, (const float *)u,
                           mean, rstd, (const float *)bias, (float *)y, (int)n);
    return ob_launch_status("normalize_rows");
}

// --------------------------------------------------------------- decode step --

int ob_cu_count()
{
    static int cus[OB_MAX_DEVICES] = {};
    const int dev = ob_device_index();
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// the kernels read the biases of a checkpoint with config.attention_bias as 16-byte vectors (advisor, round 5)
static int ob_check_bias_align(const onebit_layer_t &L, const char *fn, int l)
{
    if ((L.q_bias && !ob_aligned(L.q_bias, 16)) || (L.k_bias && !ob_aligned(L.k_bias, 16)) || (L.v_bias && !ob_aligned(L.v_bias, 16)) ||
        (L.o_bias && !ob_aligned(L.o_bias, 16)))
        return ob_fail(ONEBIT_E_ALIGN, "%s: the biases of layer %d must be 16-byte aligned", fn, l);
    return 0;
}

static int ob_fill_proj(ObProj &d, const onebit_proj_t &s, void *u, const char *name, float *st = nullptr)
{
    d.st = (s.N % 16 == 0) ? st : nullptr;      // tile partials exist for whole 16-row tiles only
    if (!s.weight || !s.input_factor || !s.weight_scale || !u)
        return ob_fail(ONEBIT_E_ARG, "decode_step: null pointer in projection %s", name);
    if (s.K % 32 != 0 || s.ldw_bytes % 4 != 0 || s.ldw_bytes < s.K / 8 || s.N <= 0)
        return ob_fail(ONEBIT_E_SHAPE, "decode_step: projection %s needs K %% 32 == 0 and 4-byte aligned rows", name);
    if (s.K > 16384) return ob_fail(ONEBIT_E_SHAPE, "decode_step: projection %s: K > 16384 unsupported", name);
    d.w = (const uint32_t *)s.weight;
    d.h = (const _Float16 *)s.input_factor;
    d.g = (const _Float16 *)s.weight_scale;
    d.u = (_Float16 *)u;
    d.N = (int)s.N; d.K = (int)s.K; d.ldw = (int)(s.ldw_bytes / 4);
    return 0;
}

template <int KV, int MS, bool ALIGNED, int PRO, int MATH, int NPROJ, bool PST, bool WGP = false, bool BIAS = false, bool ZOUT = false>
static void ob_launch_dec_gemv_t2(const ObGemvArgs &a, int G, size_t lds, hipStream_t s)
{
    static bool attr_set[OB_MAX_DEVICES] = {};
    ob_set_max_lds_once(ob_dec_gemv_kernel<KV, MS, ALIGNED, PRO, MATH, NPROJ, PST, WGP, BIAS, ZOUT>, attr_set, 160 * 1024);
    hipLaunchKernelGGL((ob_dec_gemv_kernel<KV, MS, ALIGNED, PRO, MATH, NPROJ, PST, WGP, BIAS, ZOUT>), dim3(G), dim3(OB_DEC_THREADS), lds, s, a);
}

// PST (statistics from the producers' tile partials) exists for the prologues that normalise an
// input vector; chosen when the caller supplied the partials of every such vector
template <int KV, int MS, bool ALIGNED, int PRO, int MATH, int NPROJ>
static void ob_launch_dec_gemv_t(const ObGemvArgs &a, int G, size_t lds, hipStream_t s)
{
    if constexpr (PRO == OB_P_RES_LN_RMS) {
        if (a.st_prev) return ob_launch_dec_gemv_t2<KV, MS, ALIGNED, PRO, MATH, NPROJ, true>(a, G, lds, s);
    }
    if constexpr (PRO == OB_P_SWIGLU) {
        if (a.st_gate && a.st_up) return ob_launch_dec_gemv_t2<KV, MS, ALIGNED, PRO, MATH, NPROJ, true>(a, G, lds, s);
    }
    ob_launch_dec_gemv_t2<KV, MS, ALIGNED, PRO, MATH, NPROJ, false>(a, G, lds, s);
}

// the (prologue, projection count) pairs a decoder layer needs; other pairs are not instantiated
template <int KV, int MS, bool ALIGNED, int MATH>
static bool ob_launch_dec_gemv_p(const ObGemvArgs &a, int G, size_t lds, hipStream_t s)
{
    if (a.zout) {                        // fp32 partial sums of a K slice: integer-path PLAIN instances only
        if constexpr (MATH == 1 && ALIGNED) {
            if (a.prologue == OB_P_PLAIN && a.nproj == 1) { ob_launch_dec_gemv_t2<KV, MS, true, OB_P_PLAIN, 1, 1, false, false, false, true>(a, G, lds, s); return true; }
        }
        return false;
    }
    if (a.prologue == OB_P_PLAIN && a.nproj == 1) ob_launch_dec_gemv_t<KV, MS, ALIGNED, OB_P_PLAIN, MATH, 1>(a, G, lds, s);
    else if (a.prologue == OB_P_SWIGLU && a.nproj == 1) ob_launch_dec_gemv_t<KV, MS, ALIGNED, OB_P_SWIGLU, MATH, 1>(a, G, lds, s);
    else if (a.prologue == OB_P_RES_LN_RMS && a.nproj == 2) ob_launch_dec_gemv_t<KV, MS, ALIGNED, OB_P_RES_LN_RMS, MATH, 2>(a, G, lds, s);
    else if (a.prologue == OB_P_RES_LN_RMS && a.nproj == 3) ob_launch_dec_gemv_t<KV, MS, ALIGNED, OB_P_RES_LN_RMS, MATH, 3>(a, G, lds, s);
    else if (a.prologue == OB_P_EMBED_RMS && a.nproj == 3) ob_launch_dec_gemv_t<KV, MS, ALIGNED, OB_P_EMBED_RMS, MATH, 3>(a, G, lds, s);
    else return false;
    return true;
}

// OB_DECODE_MATH=f16 selects the fp16 sign-expansion kernels for A/B measurements; default is the
// integer path wherever its alignment requirement holds.
static int ob_decode_math()
{
    static int m = -1;
    if (m < 0) {
        const char *e = getenv("OB_DECODE_MATH");
        m = (e && e[0] == 'f') ? 0 : 1;
    }
    return m;
}

static int ob_ablate_mode()
{
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("OB_ABLATE");
        mode = e ? atoi(e) : 0;
        if (mode) fprintf(stderr, "[onebit] OB_ABLATE=%d (profiling mode, results are garbage)\n", mode);
    }
    return mode;
}

#ifdef OB_PROFILE_STAMPS
extern "C" int onebit_debug_read_timing(unsigned long long *host_out, int nblocks)
{
    if (!g_dbg) return -1;
    (void)hipDeviceSynchronize();
    return (int)hipMemcpy(host_out, g_dbg, (size_t)nblocks * 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif

// grid and tile slots of a decode GEMV launch: workgroup b of G owns tiles b, b + G, ... (MS slots) of every projection
static void ob_dec_gemv_geometry(const ObGemvArgs &a, int &max_tiles, int &KV, bool &aligned, int &G, int &MS)
{
    max_tiles = 0;
    for (int p = 0; p < a.nproj; ++p) max_tiles = std::max(max_tiles, (a.p[p].N + 15) / 16);
    const int Kpad = (a.K + 511) & ~511;
    const int nchunks = Kpad / 512;
    const int kv_need = (nchunks + OB_DEC_WAVES - 1) / OB_DEC_WAVES;       // = ceil(K / 4096)
    KV = kv_need <= 4 ? kv_need : 4;
    aligned = a.K % 128 == 0;
    for (int p = 0; p < a.nproj; ++p) aligned = aligned && (a.p[p].ldw % 4 == 0) && ob_aligned(a.p[p].w, 16);
    // slots per projection: instantiated 1 2 3 4 (KV = 1), 1 2 (KV = 2), 1 (KV = 4); the grid grows beyond
    // one workgroup per CU when a projection has more tiles than that
    // (A/B, round 4: OB_DEC_SWIGLU_WGS=128 runs down_proj as 128 workgroups of two tiles -- half the redundant reads of the
    //  44-66 KB of prologue vectors through the L2s, each B operand used twice)
    static const int sw_wgs = getenv("OB_DEC_SWIGLU_WGS") ? atoi(getenv("OB_DEC_SWIGLU_WGS")) : 0;
    const bool sw2 = sw_wgs > 0 && a.nproj == 1 && a.prologue == OB_P_SWIGLU && KV >= 3;
    const int ms_max = KV == 1 ? 4 : (KV == 2 || sw2 ? 2 : 1);      // KV 3, 4: one slot
    static const int wgs_per_cu = getenv("OB_DEC_WGS_PER_CU") ? atoi(getenv("OB_DEC_WGS_PER_CU")) : 1;   // A/B switch
    G = ob_cu_count() * (wgs_per_cu > 0 ? wgs_per_cu : 1);
    if (sw2 && sw_wgs < G) G = sw_wgs;
    if (max_tiles < G) G = max_tiles;
    if ((max_tiles + G - 1) / G > ms_max) G = (max_tiles + ms_max - 1) / ms_max;
    MS = (max_tiles + G - 1) / G;
}

// The packed rows a decode GEMV launch will read, as a prefetch plan for a launch BEFORE it (ob_common.h): consuming
// workgroup b reads, per projection and slot s, the 16 rows of tile s * G + b.
static ObPfPlan ob_dec_gemv_plan(const ObGemvArgs &a)
{
    ObPfPlan P = {};
    int max_tiles, KV, G, MS;
    bool aligned;
    ob_dec_gemv_geometry(a, max_tiles, KV, aligned, G, MS);
    if (!aligned || G <= 0 || G > ob_cu_count()) return P;          // (the XCD of consuming workgroup b is b & 7 only within one round)
    for (int p = 0; p < a.nproj; ++p)
        for (int sl = 0; sl < MS && P.nseg < 4; ++sl) {
            const int r0 = sl * G * 16;
            if (r0 >= a.p[p].N) break;
            P.s[P.nseg++] = {(const char *)(a.p[p].w + (size_t)r0 * a.p[p].ldw), (long long)a.p[p].ldw * 4, a.K / 8, 16, a.p[p].N - r0,
                             0, std::min(G, (a.p[p].N - r0 + 15) / 16)};
        }
    return P;
}

// One projection per workgroup (ob_decode.h, WGP): the smallest slot count MS <= 8 for which the projections' workgroups
// (ceil(tiles_p / MS) each) fit one round of CUs.  7B: q|k|v 3 x 64 workgroups of 4 tiles, gate|up 2 x 115 of 6;
// 13B: 3 x 80 of 4, 2 x 124 of 7.
static bool ob_dec_wgp_geometry(const ObGemvArgs &a, int &MS, int (&wg_end)[3])
{
    for (MS = 1; MS <= 8; ++MS) {
        int tot = 0;
        for (int p = 0; p < 3; ++p) {
            if (p < a.nproj) tot += ((a.p[p].N + 15) / 16 + MS - 1) / MS;
            wg_end[p] = tot;
        }
        if (tot <= ob_cu_count()) return true;
    }
    return false;
}

template <int KV, int MS>
static bool ob_launch_dec_gemv_wgp(const ObGemvArgs &a, int G, size_t lds, hipStream_t s)
{
    if (a.zout) {
        if (a.prologue != OB_P_PLAIN) return false;
        ob_launch_dec_gemv_t2<KV, MS, true, OB_P_PLAIN, 1, 1, false, true, false, true>(a, G, lds, s);
        return true;
    }
    if (a.prologue == OB_P_RES_LN_RMS && a.st_prev && a.bias_prev) ob_launch_dec_gemv_t2<KV, MS, true, OB_P_RES_LN_RMS, 1, 1, true, true, true>(a, G, lds, s);
    else if (a.bias_prev) return false;
    else if (a.prologue == OB_P_RES_LN_RMS && a.st_prev) ob_launch_dec_gemv_t2<KV, MS, true, OB_P_RES_LN_RMS, 1, 1, true, true>(a, G, lds, s);
    else if (a.prologue == OB_P_RES_LN_RMS) ob_launch_dec_gemv_t2<KV, MS, true, OB_P_RES_LN_RMS, 1, 1, false, true>(a, G, lds, s);
    else if (a.prologue == OB_P_EMBED_RMS) ob_launch_dec_gemv_t2<KV, MS, true, OB_P_EMBED_RMS, 1, 1, false, true>(a, G, lds, s);
    else return false;
    return true;
}

static int ob_launch_dec_gemv(const ObGemvArgs &a_in, hipStream_t s)
{
    ObGemvArgs a = a_in;
    a.ablate = ob_ablate_mode();
#ifdef OB_PROFILE_STAMPS
    a.dbg = ob_dbg_buffer();
#endif
    int max_tiles, KV, G, MS;
    bool aligned;
    ob_dec_gemv_geometry(a, max_tiles, KV, aligned, G, MS);
    const int Kpad = (a.K + 511) & ~511;
    // launches of several projections: one projection per workgroup (OB_DEC_WGP=0: the per-slot form, A/B)
    static const int wgp_env = getenv("OB_DEC_WGP") ? atoi(getenv("OB_DEC_WGP")) : 1;
    if (wgp_env && a.nproj >= 2 && aligned && ob_decode_math() == 1 && KV <= 2 &&
        (a.prologue == OB_P_RES_LN_RMS || a.prologue == OB_P_EMBED_RMS || (a.prologue == OB_P_PLAIN && a.zout))) {
        int MSw;
        if (ob_dec_wgp_geometry(a, MSw, a.wg_end)) {
            const int Gw = a.wg_end[a.nproj - 1];
            const size_t lds_w = ob_dec_lds_i8_bytes(1, KV, MSw);        // (WGP: one projection per workgroup, MT = MS)
            bool hit = lds_w > 160 * 1024, ok = false;        // (cannot happen for K <= 8192; then the per-slot form below)
#define OB_WCASE(P, M) if (!hit && KV == P && MSw == M) { hit = true; ok = ob_launch_dec_gemv_wgp<P, M>(a, Gw, lds_w, s); }
            OB_WCASE(1, 1) OB_WCASE(1, 2) OB_WCASE(1, 3) OB_WCASE(1, 4) OB_WCASE(1, 5) OB_WCASE(1, 6) OB_WCASE(1, 7) OB_WCASE(1, 8)
            OB_WCASE(2, 1) OB_WCASE(2, 2) OB_WCASE(2, 3) OB_WCASE(2, 4) OB_WCASE(2, 5) OB_WCASE(2, 6) OB_WCASE(2, 7) OB_WCASE(2, 8)
#undef OB_WCASE
            if (hit && ok) return ob_launch_status("decode gemv");
        }
    }
    // the producer's bias (o_proj with config.attention_bias) exists in the one-projection-per-workgroup kernels fed with tile
    // partials -- what onebit_decode_step launches for every LLaMA shape; no other instance carries it
    if (a.bias_prev)
        return ob_fail(ONEBIT_E_SHAPE, "decode gemv: o_proj bias needs the WGP kernels (>= 2 projections, K %% 128 == 0, K <= 8192, 16-byte "
                                       "aligned rows) and the producer's tile partials (state->tile_stats, N %% 16 == 0); decode this "
                                       "checkpoint through the module path");
    if (!aligned && KV != 1)
        return ob_fail(ONEBIT_E_SHAPE, "decode gemv: in_features > 4096 needs K %% 128 == 0 and 16-byte aligned rows");
    const int MT = MS * a.nproj;
    // (integer path: the digit image covers all KV * 8 chunks of a wave row, ob_decode.h)
    const size_t lds_i8 = ob_dec_lds_i8_bytes(a.nproj, KV, MT);
    // the integer path pays a per-projection quantisation; with one 512-weight chunk per wave it does not pay back
    // single-chunk launches (o_proj: one tile, one 512-weight chunk per wave) take the integer path as well since round 4
    // (8 MFMAs + 32 v_and instead of 16 MFMAs + 227 sign-expansion instructions: 3.54 -> 3.07 us per launch in a chain;
    //  in rounds 1-2 the per-wave quantisation did not pay back).  OB_DEC_I8_SINGLE=0: A/B.
    static const int i8_single = getenv("OB_DEC_I8_SINGLE") ? atoi(getenv("OB_DEC_I8_SINGLE")) : 1;
    const bool use_i8 = aligned && ob_decode_math() == 1 && lds_i8 <= 160 * 1024 && (MT * KV >= 2 || i8_single);
    const size_t lds = use_i8 ? lds_i8 : (size_t)a.nproj * Kpad * 2 + (size_t)MT * OB_DEC_WAVES * 16 * 4 + 256 * 4;
    if (lds > 160 * 1024) return ob_fail(ONEBIT_E_SHAPE, "decode gemv: LDS need %zu > 160 KiB", lds);
    bool ok = false, hit = false;
#define OB_CASE(P, M)                                                              \
    if (!hit && aligned && KV == P && MS == M) {                                   \
        hit = true;                                                                \
        ok = use_i8 ? ob_launch_dec_gemv_p<P, M, true, 1>(a, G, lds, s)            \
                    : ob_launch_dec_gemv_p<P, M, true, 0>(a, G, lds, s);           \
    }
#define OB_CASE_U(M)                                                               \
    if (!hit && !aligned && MS == M) {                                             \
        hit = true;                                                                \
        ok = ob_launch_dec_gemv_p<1, M, false, 0>(a, G, lds, s);                   \
    }
    OB_CASE(1, 1) OB_CASE(1, 2) OB_CASE(1, 3) OB_CASE(1, 4)
    OB_CASE(2, 1) OB_CASE(2, 2)
    OB_CASE(3, 1) OB_CASE(3, 2)
    OB_CASE(4, 1) OB_CASE(4, 2)
    OB_CASE_U(1) OB_CASE_U(2) OB_CASE_U(3) OB_CASE_U(4)
#undef OB_CASE
#undef OB_CASE_U
    if (!hit) return ob_fail(ONEBIT_E_SHAPE, "decode gemv: no kernel instance for KV=%d MS=%d", KV, MS);
    if (!ok && a.zout)
        return ob_fail(ONEBIT_E_SHAPE, "decode gemv: the fp32-partial (K-sharded) form needs a K slice that is a multiple of 128 with 16-byte "
                                       "aligned rows, the integer path, and 1 projection (or 2-3 with K <= 8192): prologue %d, %d projections, K = %d",
                       a.prologue, a.nproj, a.K);
    if (!ok) return ob_fail(ONEBIT_E_FLAG, "decode gemv: prologue %d with %d projections is not instantiated "
                            "(PLAIN:1, SWIGLU:1, RES_LN_RMS:2|3, EMBED_RMS:3)", a.prologue, a.nproj);
    return ob_launch_status("decode gemv");
}

static int ob_single_token_gemv(const void *packed, int64_t ldw_bytes, const void *x, const void *h, const void *g,
                                void *u, int64_t K, int64_t N, hipStream_t s)
{
    ObGemvArgs a = {};
    a.nproj = 1; a.K = (int)K; a.prologue = OB_P_PLAIN;
    a.p[0].w = (const uint32_t *)packed; a.p[0].h = (const _Float16 *)h; a.p[0].g = (const _Float16 *)g;
    a.p[0].u = (_Float16 *)u; a.p[0].N = (int)N; a.p[0].K = (int)K; a.p[0].ldw = (int)(ldw_bytes / 4);
    a.xin = (const _Float16 *)x;
    a.rms_eps = 1e-6f; a.ln_eps = 1e-5f;
    return ob_launch_dec_gemv(a, s);
}

extern "C" int onebit_rows_res_ln_rms_bias(const void *hres_in, const void *u_prev, const void *bias_prev, const void *rms_w, void *hres_out,
                                           void *x, const void *const *h_next, void *const *x_scaled, int32_t n_scaled,
                                           int64_t T, int64_t H, float rms_eps, float ln_eps, void *stream)
{
    if (T < 0 || H <= 0 || n_scaled < 0 || n_scaled > 3) return ob_fail(ONEBIT_E_ARG, "rows_res_ln_rms: bad size");
    if (H % 8 != 0 || H > OB_DEC_MAXV * OB_DEC_THREADS * 8) return ob_fail(ONEBIT_E_SHAPE, "rows_res_ln_rms: H = %lld", (long long)H);
    if (T == 0) return 0;
    if (!hres_in || !u_prev || !rms_w || !hres_out || (!x && n_scaled == 0) || (n_scaled > 0 && (!h_next || !x_scaled)))
        return ob_fail(ONEBIT_E_ARG, "rows_res_ln_rms: null pointer");
    if (T > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "rows_res_ln_rms: dimension too large");
    if (bias_prev && !ob_aligned(bias_prev, 16)) return ob_fail(ONEBIT_E_ALIGN, "rows_res_ln_rms: bias must be 16-byte aligned");
    ObBNormArgs a = {};
    a.hres_in = (const _Float16 *)hres_in; a.u_prev = (const _Float16 *)u_prev; a.rms_w = (const _Float16 *)rms_w;
    a.hres_out = (_Float16 *)hres_out; a.x = (_Float16 *)x; a.H = (int)H; a.rms_eps = rms_eps; a.ln_eps = ln_eps;
    a.n_scaled = n_scaled; a.bias_prev = (const _Float16 *)bias_prev;
    for (int i = 0; i < n_scaled; ++i) {
        if (!h_next[i] || !x_scaled[i]) return ob_fail(ONEBIT_E_ARG, "rows_res_ln_rms: null scaled output %d", i);
        a.h_next[i] = (const _Float16 *)h_next[i]; a.x_scaled[i] = (_Float16 *)x_scaled[i];
    }
    OB_LAUNCH_NORM(false, H, dim3((unsigned)T), (hipStream_t)stream, a);
    return ob_launch_status("rows_res_ln_rms");
}

extern "C" int onebit_rows_res_ln_rms(const void *hres_in, const void *u_prev, const void *rms_w, void *hres_out,
                                      void *x, const void *const *h_next, void *const *x_scaled, int32_t n_scaled,
                                      int64_t T, int64_t H, float rms_eps, float ln_eps, void *stream)
{
    return onebit_rows_res_ln_rms_bias(hres_in, u_prev, nullptr, rms_w, hres_out, x, h_next, x_scaled, n_scaled, T, H, rms_eps, ln_eps, stream);
}

extern "C" int onebit_rows_swiglu_stats(const void *u_gate, const void *u_up, const void *h_next, const float *row_stats, void *act,
                                        int64_t T, int64_t I, float ln_eps, void *stream);
extern "C" int onebit_rows_swiglu(const void *u_gate, const void *u_up, const void *h_next, void *act, int64_t T, int64_t I,
                                  float ln_eps, void *stream)
{
    return onebit_rows_swiglu_stats(u_gate, u_up, h_next, nullptr, act, T, I, ln_eps, stream);
}

extern "C" int onebit_rows_swiglu_stats(const void *u_gate, const void *u_up, const void *h_next, const float *row_stats, void *act,
                                        int64_t T, int64_t I, float ln_eps, void *stream)
{
    if (T < 0 || I <= 0) return ob_fail(ONEBIT_E_ARG, "rows_swiglu: bad size");
    if (I % 8 != 0 || I > OB_DEC_MAXV * OB_DEC_THREADS * 8) return ob_fail(ONEBIT_E_SHAPE, "rows_swiglu: I = %lld", (long long)I);
    if (T == 0) return 0;
    if (!u_gate || !u_up || !act) return ob_fail(ONEBIT_E_ARG, "rows_swiglu: null pointer");
    if (T > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "rows_swiglu: dimension too large");
    if (row_stats && !ob_aligned(row_stats, 16)) return ob_fail(ONEBIT_E_ALIGN, "rows_swiglu: row_stats must be 16-byte aligned");
    ObBSwigluArgs a = {(const _Float16 *)u_gate, (const _Float16 *)u_up, (_Float16 *)act, (int)I, ln_eps, (const _Float16 *)h_next, row_stats};
    OB_LAUNCH_SWIGLU(I, dim3((unsigned)T), (hipStream_t)stream, a);
    return ob_launch_status("rows_swiglu");
}

// 1 when a call of this shape may pass ONEBIT_FLAG_PRESCALED (it would take the LDS-DMA GEMM, which consumes
// pre-scaled rows); the other kernels multiply by h on the way in and cannot skip it
extern "C" int onebit_linear_prescaled_ok(int64_t T, int64_t K, int64_t N, int dtype)
{
    if (dtype != ONEBIT_F16 || T <= 0 || K <= 0 || N <= 0 || K % 32 != 0) return 0;
    if (ob_gemm3_ok(T, K, N)) return 1;
    // 2 <= T <= 64: the LDS-DMA skinny GEMM (ob_skinny3.h) consumes pre-scaled rows too
    return (T >= 2 && T <= 64 && K % 128 == 0 && K >= 512 && T * K * 2 < ((int64_t)1 << 32)) ? 1 : 0;
}

extern "C" int onebit_rows_qkv_rope_stats(const void *u_q, const void *u_k, const void *u_v, const void *cos, const void *sin,
                                          const float *row_stats, void *q, void *k_cache, void *v_cache, int64_t B, int64_t S,
                                          int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t past_len, int64_t max_len,
                                          int64_t max_pos, float ln_eps, unsigned flags, void *stream);
extern "C" int onebit_rows_qkv_rope(const void *u_q, const void *u_k, const void *u_v, const void *cos, const void *sin,
                                    void *q, void *k_cache, void *v_cache, int64_t B, int64_t S, int32_t n_heads,
                                    int32_t n_kv_heads, int32_t head_dim, int64_t past_len, int64_t max_len, int64_t max_pos,
                                    float ln_eps, unsigned flags, void *stream)
{
    return onebit_rows_qkv_rope_stats(u_q, u_k, u_v, cos, sin, nullptr, q, k_cache, v_cache, B, S, n_heads, n_kv_heads, head_dim,
                                      past_len, max_len, max_pos, ln_eps, flags, stream);
}

extern "C" int onebit_rows_qkv_rope_stats(const void *u_q, const void *u_k, const void *u_v, const void *cos, const void *sin,
                                          const float *row_stats, void *q, void *k_cache, void *v_cache, int64_t B, int64_t S,
                                          int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t past_len, int64_t max_len,
                                          int64_t max_pos, float ln_eps, unsigned flags, void *stream)
{
    if (flags & ~ONEBIT_FLAG_Q_TOKEN_MAJOR) return ob_fail(ONEBIT_E_FLAG, "rows_qkv_rope: unknown flags 0x%x", flags);
    if (B < 0 || S < 0 || n_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0 || past_len < 0)
        return ob_fail(ONEBIT_E_ARG, "rows_qkv_rope: bad size");
    // (power-of-two head_dim: a head is an aligned group of head_dim / 8 lanes, rotate_half partners are a lane exchange)
    if (head_dim < 16 || (head_dim & (head_dim - 1)) != 0 || (int64_t)n_heads * head_dim > OB_DEC_MAXV * OB_DEC_THREADS * 8 || n_kv_heads > n_heads)
        return ob_fail(ONEBIT_E_SHAPE, "rows_qkv_rope: heads %d / %d x %d", n_heads, n_kv_heads, head_dim);
    if (past_len + S > max_len || past_len + S > max_pos)
        return ob_fail(ONEBIT_E_SHAPE, "rows_qkv_rope: %lld + %lld tokens beyond the cache (%lld) or the rope tables (%lld)",
                       (long long)past_len, (long long)S, (long long)max_len, (long long)max_pos);
    if (B == 0 || S == 0) return 0;
    if (!u_q || !u_k || !u_v || !cos || !sin || !q || !k_cache || !v_cache) return ob_fail(ONEBIT_E_ARG, "rows_qkv_rope: null pointer");
    if (B * S > 0x7fffffffLL || max_len > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "rows_qkv_rope: dimension too large");
    ObQkvRopeArgs a = {(const _Float16 *)u_q, (const _Float16 *)u_k, (const _Float16 *)u_v, (const _Float16 *)cos, (const _Float16 *)sin,
                       (_Float16 *)q, (_Float16 *)k_cache, (_Float16 *)v_cache, (int)S, n_heads, n_kv_heads, head_dim, (int)past_len,
                       (int)max_len, (flags & ONEBIT_FLAG_Q_TOKEN_MAJOR) ? 1 : 0, ln_eps, row_stats};
    OB_LAUNCH_QKVROPE((int64_t)n_heads * head_dim, dim3((unsigned)(B * S)), (hipStream_t)stream, a);
    return ob_launch_status("rows_qkv_rope");
}

// Ragged rows (ABI 9): LayerNorm + RoPE + cache append for the token rows of SEVERAL sequences -- row t -> (row_slot[t], row_pos[t])
extern "C" int onebit_rows_qkv_rope_ragged(const void *u_q, const void *u_k, const void *u_v, const void *cos, const void *sin,
                                           const int32_t *row_slot, const int32_t *row_pos, void *q, void *k_cache, void *v_cache,
                                           const void *q_bias, const void *k_bias, const void *v_bias, int64_t T, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t n_slots,
                                           int64_t max_len, int64_t max_pos, float ln_eps, void *stream)
{
    if (T < 0 || n_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0 || n_slots <= 0 || max_len <= 0) return ob_fail(ONEBIT_E_ARG, "rows_qkv_rope_ragged: bad size");
    if (head_dim < 16 || (head_dim & (head_dim - 1)) != 0 || (int64_t)n_heads * head_dim > OB_DEC_MAXV * OB_DEC_THREADS * 8 || n_kv_heads > n_heads)
        return ob_fail(ONEBIT_E_SHAPE, "rows_qkv_rope_ragged: heads %d / %d x %d", n_heads, n_kv_heads, head_dim);
    // positions are device-side: the kernel skips rows outside [0, max_len); the rope tables must cover every cache position
    if (max_len > max_pos) return ob_fail(ONEBIT_E_SHAPE, "rows_qkv_rope_ragged: cache rows (%lld) beyond the rope tables (%lld)", (long long)max_len, (long long)max_pos);
    if (T == 0) return 0;
    if (!u_q || !u_k || !u_v || !cos || !sin || !q || !k_cache || !v_cache || !row_pos) return ob_fail(ONEBIT_E_ARG, "rows_qkv_rope_ragged: null pointer");
    if (!ob_aligned(u_q, 16) || !ob_aligned(u_k, 16) || !ob_aligned(u_v, 16) || !ob_aligned(q, 16) || !ob_aligned(k_cache, 16) || !ob_aligned(v_cache, 16) ||
        !ob_aligned(cos, 16) || !ob_aligned(sin, 16))
        return ob_fail(ONEBIT_E_ALIGN, "rows_qkv_rope_ragged: tensors must be 16-byte aligned");
    if (T > 0x7fffffffLL || max_len > 0x7fffffffLL || n_slots > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "rows_qkv_rope_ragged: dimension too large");
    if ((q_bias || k_bias || v_bias) && !(q_bias && k_bias && v_bias)) return ob_fail(ONEBIT_E_ARG, "rows_qkv_rope_ragged: some but not all of q_bias / k_bias / v_bias");
    if (q_bias && (!ob_aligned(q_bias, 16) || !ob_aligned(k_bias, 16) || !ob_aligned(v_bias, 16)))
        return ob_fail(ONEBIT_E_ALIGN, "rows_qkv_rope_ragged: biases must be 16-byte aligned");
    ObQkvRopeArgs a = {(const _Float16 *)u_q, (const _Float16 *)u_k, (const _Float16 *)u_v, (const _Float16 *)cos, (const _Float16 *)sin,
                       (_Float16 *)q, (_Float16 *)k_cache, (_Float16 *)v_cache, 1, n_heads, n_kv_heads, head_dim, 0,
                       (int)max_len, 1, ln_eps, nullptr, row_slot, row_pos, (int)n_slots,
                       (const _Float16 *)q_bias, (const _Float16 *)k_bias, (const _Float16 *)v_bias};
    OB_LAUNCH_QKVROPE((int64_t)n_heads * head_dim, dim3((unsigned)T), (hipStream_t)stream, a);
    return ob_launch_status("rows_qkv_rope_ragged");
}

// Tile partials: one slot per pre-LayerNorm vector of a layer (q, k, v, o, gate, up, down), each
// 2 floats per 16-row tile, padded to whole blocks of 256 tiles (the consumers read whole blocks).
struct ObStatsLayout { size_t off[7]; size_t total; };
static ObStatsLayout ob_stats_layout(const onebit_model_t *m)
{
    const int64_t n[7] = {(int64_t)m->n_heads * m->head_dim, (int64_t)m->n_kv_heads * m->head_dim,
                          (int64_t)m->n_kv_heads * m->head_dim, m->hidden, m->intermediate, m->intermediate, m->hidden};
    ObStatsLayout sl;
    size_t o = 0;
    for (int i = 0; i < 7; ++i) {
        sl.off[i] = o;
        o += (size_t)((n[i] + 4095) / 4096) * 512;
    }
    sl.total = o;
    return sl;
}

extern "C" size_t onebit_decode_stats_floats(const onebit_model_t *m)
{
    if (!m || m->n_heads <= 0 || m->n_kv_heads <= 0 || m->head_dim <= 0 || m->hidden <= 0 || m->intermediate <= 0) return 0;
    return ob_stats_layout(m).total;
}

extern "C" size_t onebit_batch_stats_floats(const onebit_model_t *m, int32_t batch)
{
    if (!m || batch <= 0) return 0;
    const int NQ = m->n_heads * m->head_dim, NK = m->n_kv_heads * m->head_dim;
    return (size_t)batch * ((size_t)ob_tile_stats_floats(NQ) + 2 * (size_t)ob_tile_stats_floats(NK));
}

extern "C" size_t onebit_attn_scratch_bytes(const onebit_model_t *m, int32_t S)
{
    if (!m || S < 2 || m->n_heads <= 0 || m->max_len <= 0) return 0;
    return (size_t)m->n_heads * ((size_t)m->max_len * 4 + (size_t)S * 2 * 4 + (size_t)S * 128 * 4 + 4) + 64;
}

// ---- skinny GEMM, LDS-DMA form (ob_skinny3.h): pre-scaled rows, 2 <= T <= 64, 16 * rnt rows per 4-wave workgroup
template <bool PARTIAL, int RT, int RNT, int NW>
static void ob_launch_sk3_t(const ObSk3Args &a_in, int grid, hipStream_t s)
{
    ObSk3Args a = a_in;
#ifdef OB_PROFILE_STAMPS
    a.dbg = ob_dbg_buffer();
#endif
    // ring depth per wave: what fits 160 KB of LDS with NW private rings (a piece is 5 / 9 / 17 KB at 16 / 32 / 64 tokens)
    constexpr int NBUF = RT == 1 ? 3 : 2;
    constexpr int lds = OB_SK3_LDS(RT, RNT, NBUF, NW);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set[OB_MAX_DEVICES] = {};
    ob_set_max_lds_once(ob_skinny3_kernel<PARTIAL, RT, RNT, NBUF, NW>, attr_set, lds);
    hipLaunchKernelGGL((ob_skinny3_kernel<PARTIAL, RT, RNT, NBUF, NW>), dim3(grid), dim3(64 * NW), lds, s, a);
}
static bool ob_skinny3_shape_ok(const void *packed, int64_t ldw_bytes, const void *a, int64_t lda, int64_t T, int64_t K, int64_t N)
{
    return T >= 2 && T <= 64 && K % 128 == 0 && K >= 512 && ldw_bytes % 16 == 0 && ob_aligned(packed, 16) && ob_aligned(a, 16) &&
           lda % 8 == 0 && T * lda * 2 < ((int64_t)1 << 32) && N * ldw_bytes < ((int64_t)1 << 32);
}
// 16-row tiles per workgroup: the smallest count whose grid fits one round of workgroups (one 144 KB workgroup per CU);
// `copies` = K-slices per projection (split-K launches)
static int ob_skinny3_pick_rnt(const int64_t *N, int np, int copies)
{
    static const int env = getenv("OB_SK3_RNT") ? atoi(getenv("OB_SK3_RNT")) : 0;      // A/B switch
    if (env == 1 || env == 2 || env == 3 || env == 4 || env == 6 || env == 8) return env;
    static const int cand[6] = {1, 2, 3, 4, 6, 8};
    for (int c = 0; c < 6; ++c) {
        int64_t wgs = 0;
        for (int i = 0; i < np; ++i) wgs += (N[i] + 16 * cand[c] - 1) / (16 * cand[c]);
        if (wgs * copies <= ob_cu_count()) return cand[c];
    }
    return 4;
}
static int ob_skinny3_rnt(int64_t N) { return ob_skinny3_pick_rnt(&N, 1, 1); }
// 8 waves (K split eight ways, two waves per SIMD: one wave's transfers wait while the other multiplies) wherever the
// private rings fit the LDS; 64-token tiles: 4 waves.  (4-wave workgroups at 16 / 32 tokens were measured and are not built:
// gate|up 8.9 vs 8.2 us per launch, the K-sliced o / down launches equal.)
template <bool PARTIAL, int RT, int RNT>
static void ob_launch_sk3_nw(const ObSk3Args &a, int grid, hipStream_t s)
{
    ob_launch_sk3_t<PARTIAL, RT, RNT, RT == 4 ? 4 : 8>(a, grid, s);
}
template <bool PARTIAL>
static bool ob_launch_skinny3(const ObSk3Args &a, int grid, int rnt, hipStream_t s)
{
    const int RTv = a.T <= 16 ? 1 : (a.T <= 32 ? 2 : 4);
    bool hit = false;
#define OB_SK3(RT_, RNT_)                                                                 \
    if (!hit && RTv == RT_ && rnt == RNT_) {                                              \
        hit = true;                                                                       \
        ob_launch_sk3_nw<PARTIAL, RT_, RNT_>(a, grid, s);                                 \
    }
    OB_SK3(1, 1) OB_SK3(1, 2) OB_SK3(1, 3) OB_SK3(1, 4) OB_SK3(1, 6) OB_SK3(1, 8)
    OB_SK3(2, 1) OB_SK3(2, 2) OB_SK3(2, 3) OB_SK3(2, 4) OB_SK3(2, 6) OB_SK3(2, 8)
    OB_SK3(4, 1) OB_SK3(4, 2) OB_SK3(4, 3) OB_SK3(4, 4) OB_SK3(4, 6) OB_SK3(4, 8)
#undef OB_SK3
    return hit;
}

static int ob_batched_head(const onebit_model_t *m, const onebit_batch_state_t *st, hipStream_t s);

// The decoder layers + final norm of the batched step for the `st->batch` sequences whose first KV-cache slot is
// `slot0` (every pointer of `st` already addresses row 0 of that range; the caches are indexed from the model).
static int ob_batched_layers(const onebit_model_t *m, const onebit_batch_state_t *st, int slot0, hipStream_t s)
{
    const int B = st->batch, H = m->hidden, I = m->intermediate, D = m->head_dim;
    const int NQ = m->n_heads * D, NK = m->n_kv_heads * D;
    if (NQ != H) return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: n_heads * head_dim != hidden");
    _Float16 *hA = (_Float16 *)st->hres0, *hB = (_Float16 *)st->hres1;
    int rc;
    auto gemm = [&](const onebit_proj_t &p, const void *xin, void *uout, int64_t K, int64_t N, const char *name) -> int {
        if (!p.weight || !p.input_factor || !p.weight_scale || p.K != K || p.N != N || p.K % 32 != 0 || p.ldw_bytes % 4 != 0 ||
            p.ldw_bytes < p.K / 8)
            return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: projection %s has an unexpected shape", name);
        ob_launch_mm16<false>(p.weight, p.ldw_bytes, xin, K, p.input_factor, p.weight_scale, uout, nullptr, B, K, N, s);
        return ob_launch_status("decode_step_batched(gemm)");
    };
    // split-K for down_proj needs two fp32 [B, H] scratch rows: u_gate / u_up ([B, I] fp16) are free by then
    float *zs0 = (float *)st->u_gate, *zs1 = (float *)st->u_up;
    static const int splitk_env = getenv("OB_BATCH_SPLITK") ? atoi(getenv("OB_BATCH_SPLITK")) : 1;
    bool splitk_down = splitk_env && (int64_t)I * 2 >= (int64_t)H * 4 && I % 256 == 0 && H % 8 == 0;
    for (int l = 0; splitk_down && l < m->n_layers; ++l)
        splitk_down = ob_skinny_ok((const uint32_t *)m->layers[l].down.weight + (I / 2) / 32, m->layers[l].down.ldw_bytes, B, I / 2) &&
                      ob_skinny_ok(m->layers[l].down.weight, m->layers[l].down.ldw_bytes, B, I / 2);
    struct P3 { const onebit_proj_t *p[3]; };
    struct U3 { void *u[3]; };
    struct N3 { int64_t n[3]; };
    struct S3 { float *s[3]; };
    bool stats_written = false;
    // projections sharing their input: one skinny launch over all their row tiles, or one launch each
    auto gemm_multi = [&](P3 ps, U3 us, N3 ns, S3 ss, int np, const void *xin, int64_t K, const char *name) -> int {
        stats_written = false;
        bool fuse = true;
        for (int i = 0; i < np; ++i) {
            const onebit_proj_t &p = *ps.p[i];
            if (!p.weight || !p.input_factor || !p.weight_scale || p.K != K || p.N != ns.n[i] || p.K % 32 != 0 ||
                p.ldw_bytes % 4 != 0 || p.ldw_bytes < p.K / 8)
                return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: projection %s[%d] has an unexpected shape", name, i);
            fuse = fuse && ob_skinny_ok(p.weight, p.ldw_bytes, B, K);
        }
        if (!fuse) {
            for (int i = 0; i < np; ++i) {
                const onebit_proj_t &p = *ps.p[i];
                ob_launch_mm16<false>(p.weight, p.ldw_bytes, xin, K, p.input_factor, p.weight_scale, us.u[i], nullptr, B, K, p.N, s);
            }
            return ob_launch_status("decode_step_batched(gemm)");
        }
        // (64 rows per workgroup.  Other tile counts were measured, 32 slots: 128 rows for the wide launches 2.85 vs 2.75 ms
        //  (round 2); 48 rows for q|k|v / 96 for gate|up -- grids of 258 / 230 workgroups for 256 CUs -- 15.5 / 14.1 us against
        //  14.7 / 14.7 at 7B and slower at 13B (round 3): the launch is bound by its per-phase instruction stream and barriers,
        //  not by the grid fit)
        const int rows = 64;
        ObSkinnyArgs ka = {};
        int tiles = 0;
        for (int i = 0; i < 3; ++i) {
            const int j = i < np ? i : np - 1;
            const onebit_proj_t &p = *ps.p[j];
            if (i < np) tiles += (int)((p.N + rows - 1) / rows);
            ka.p[i] = {(const uint32_t *)p.weight, (long long)(p.ldw_bytes / 4), (const _Float16 *)p.input_factor,
                       (const _Float16 *)p.weight_scale, (const _Float16 *)xin, (_Float16 *)us.u[j], nullptr, (int)p.N,
                       (int)K, tiles, ss.s[j]};
        }
        stats_written = ss.s[0] != nullptr;
        ka.ldx = K; ka.T = B;
        if (B <= 16) ob_launch_skinny<false, 1>(ka, tiles, s);
        else if (B <= 32) ob_launch_skinny<false, 2>(ka, tiles, s);
        else ob_launch_skinny<false, 4>(ka, tiles, s);
        return ob_launch_status("decode_step_batched(gemm)");
    };
    // Short-and-wide projections (N = hidden: o, down) have 64 row tiles for 256 CUs: split K over two
    // workgroup ranges, fp32 partial sums into two free [B, H] scratch rows, summed by the next norm kernel
    auto gemm_splitk2 = [&](const onebit_proj_t &p, const void *xin, int64_t K, float *z0, float *z1, const char *name) -> int {
        if (!p.weight || !p.input_factor || !p.weight_scale || p.K != K || p.N != H || p.ldw_bytes % 16 != 0)
            return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: projection %s has an unexpected shape", name);
        const int Kh = (int)(K / 2), tiles1 = (H + 63) / 64;
        ObSkinnyArgs ka = {};
        for (int i = 0; i < 3; ++i) {
            const int j = i < 2 ? i : 1;
            ka.p[i] = {(const uint32_t *)p.weight + j * (Kh / 32), (long long)(p.ldw_bytes / 4),
                       (const _Float16 *)p.input_factor + j * Kh, (const _Float16 *)p.weight_scale,
                       (const _Float16 *)xin + j * Kh, nullptr, j == 0 ? z0 : z1, H, Kh, tiles1 * (j + 1)};
        }
        ka.ldx = K; ka.T = B;
        if (B <= 16) ob_launch_skinny<true, 1>(ka, 2 * tiles1, s);
        else if (B <= 32) ob_launch_skinny<true, 2>(ka, 2 * tiles1, s);
        else ob_launch_skinny<true, 4>(ka, 2 * tiles1, s);
        return ob_launch_status("decode_step_batched(split-K gemm)");
    };
    // o_proj: K = hidden halves; its partial sums use the u_gate / u_up rows as well (free until gate|up runs)
    bool splitk_o = splitk_env && H % 256 == 0 && (int64_t)I * 2 >= (int64_t)H * 4 && NQ == H;
    for (int l = 0; splitk_o && l < m->n_layers; ++l)
        splitk_o = m->layers[l].o.weight && m->layers[l].o.ldw_bytes % 16 == 0 &&
                   ob_skinny_ok((const uint32_t *)m->layers[l].o.weight + (H / 2) / 32, m->layers[l].o.ldw_bytes, B, H / 2) &&
                   ob_skinny_ok(m->layers[l].o.weight, m->layers[l].o.ldw_bytes, B, H / 2);
    // Every projection through the LDS-DMA skinny GEMM (ob_skinny3.h) when the state has room for the consumers' pre-scaled
    // rows (x_scaled: 3 x [B, hidden]): the row kernels write fp16(x * h_p) per consuming projection (the rounding of
    // bitnet.py:113 done by the producer), attention and SwiGLU scale their outputs for o_proj / down_proj, down_proj runs
    // as two K-slices whose fp32 partial sums the next norm kernel adds.  OB_SKINNY3=0: the first-form kernels above.
    static const int sk3_env = getenv("OB_SKINNY3") ? atoi(getenv("OB_SKINNY3")) : 1;
    _Float16 *xs[3] = {nullptr, nullptr, nullptr};
    bool sk3 = sk3_env && st->x_scaled && ob_aligned(st->x_scaled, 16) && NQ == H && H % 128 == 0 && I % 256 == 0 && (int64_t)I * 2 >= (int64_t)H * 4;
    if (sk3)
        for (int i = 0; i < 3; ++i) xs[i] = (_Float16 *)st->x_scaled + (size_t)i * B * H;
    for (int l = 0; sk3 && l < m->n_layers; ++l) {
        const onebit_layer_t &L = m->layers[l];
        const onebit_proj_t *pp[7] = {&L.q, &L.k, &L.v, &L.o, &L.gate, &L.up, &L.down};
        const int64_t kk[7] = {H, H, H, NQ, H, H, I}, nn[7] = {NQ, NK, NK, H, I, I, H};
        for (int i = 0; sk3 && i < 7; ++i) {
            const onebit_proj_t &p = *pp[i];
            // the rows each launch really DMAs from: the pre-scaled copies (q, k, v / gate, up), attn_out (o), act (down)
            const void *rows = i < 3 ? (const void *)xs[i] : i == 3 ? (const void *)st->attn_out : i < 6 ? (const void *)xs[i - 4] : (const void *)st->act;
            sk3 = p.weight && p.input_factor && p.weight_scale && p.K == kk[i] && p.N == nn[i] &&
                  ob_skinny3_shape_ok(p.weight, p.ldw_bytes, rows, kk[i], B, i == 6 ? kk[i] / 2 : kk[i], nn[i]);
        }
    }
    const bool down_parts = sk3 || splitk_down;
    struct A3 { const _Float16 *a[3]; };
    // the launch description of a skinny GEMM is built BEFORE its producer kernel is launched: the producer pulls the
    // packed rows of exactly these workgroups into the L2 of the XCD they will run on (ob_common.h, ob_prefetch_l2)
    struct Sk3Launch { ObSk3Args ka; int wgs, rnt, nseg; bool partial; };
    auto build_sk3 = [&](P3 ps, U3 us, S3 ss, A3 as, int np, int64_t K) -> Sk3Launch {
        Sk3Launch L3 = {};
        int64_t nn[3] = {0, 0, 0};
        for (int i = 0; i < np; ++i) nn[i] = ps.p[i]->N;
        L3.rnt = ob_skinny3_pick_rnt(nn, np, 1);
        for (int i = 0; i < 3; ++i) {
            const int j = i < np ? i : np - 1;
            const onebit_proj_t &p = *ps.p[j];
            if (i < np) L3.wgs += (int)((p.N + 16 * L3.rnt - 1) / (16 * L3.rnt));
            L3.ka.p[i] = {(const uint32_t *)p.weight, (long long)(p.ldw_bytes / 4), (const _Float16 *)p.weight_scale, as.a[j],
                          (_Float16 *)us.u[j], nullptr, ss.s[j], (int)p.N, (int)K, L3.wgs};
        }
        L3.ka.lda = K; L3.ka.T = B; L3.nseg = np; L3.partial = false;
        return L3;
    };
    // a projection onto the hidden width (o, down: few rows, so few workgroups) as two K-slices with fp32 partial sums in
    // zs0 / zs1, added by the next norm kernel: half the activation bytes per workgroup at the same grid (o_proj: 2.24 ->
    // 2.19-2.21 ms per 32-slot step).  FOUR slices (64-row workgroups, one or two pieces per wave: all ramp and a larger
    // reduction) measured slower: 2.23 ms at 7B, 3.56 vs 3.46 ms at 13B.
    auto build_sk3_split2 = [&](const onebit_proj_t &p, const _Float16 *a, int64_t K) -> Sk3Launch {
        Sk3Launch L3 = {};
        const int Kh = (int)(K / 2);
        const int64_t nh[1] = {p.N};
        L3.rnt = ob_skinny3_pick_rnt(nh, 1, 2);
        const int wg1 = (int)((p.N + 16 * L3.rnt - 1) / (16 * L3.rnt));
        for (int i = 0; i < 3; ++i) {
            const int j = i < 2 ? i : 1;
            L3.ka.p[i] = {(const uint32_t *)p.weight + j * (Kh / 32), (long long)(p.ldw_bytes / 4), nullptr, a + j * Kh, nullptr,
                          j == 0 ? zs0 : zs1, nullptr, (int)p.N, Kh, wg1 * (j + 1)};
        }
        L3.ka.lda = K; L3.ka.T = B; L3.wgs = 2 * wg1; L3.nseg = 2; L3.partial = true;
        return L3;
    };
    static const int sk3_pf = getenv("OB_SK3_PREFETCH") ? atoi(getenv("OB_SK3_PREFETCH")) : 1;     // A/B: producers prefetch the consumer's rows
    auto plan_of = [&](const Sk3Launch &L3) -> ObPfPlan {
        ObPfPlan P = {};
        if (!sk3_pf) return P;
        for (int i = 0; i < L3.nseg; ++i) {
            const ObSk3Proj &q = L3.ka.p[i];
            P.s[i] = {(const char *)q.W, q.ldw_words * 4, q.K / 8, 16 * L3.rnt, q.N, i ? L3.ka.p[i - 1].wg_end : 0, q.wg_end};
        }
        P.nseg = L3.nseg;
        return P;
    };
    auto launch_sk3 = [&](const Sk3Launch &L3, const char *name) -> int {
        const bool ok = L3.partial ? ob_launch_skinny3<true>(L3.ka, L3.wgs, L3.rnt, s) : ob_launch_skinny3<false>(L3.ka, L3.wgs, L3.rnt, s);
        if (!ok) return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: no skinny GEMM instance for %s", name);
        return ob_launch_status("decode_step_batched(gemm)");
    };
    static const int sk3_osplit = getenv("OB_SK3_OSPLIT") ? atoi(getenv("OB_SK3_OSPLIT")) : 1;     // A/B: o_proj as two K-slices
    const bool o_split = sk3 && sk3_osplit && NQ % 256 == 0 && NQ >= 1024;
    for (int l = 0; l < m->n_layers; ++l) {
        const onebit_layer_t &L = m->layers[l];
        if (!L.input_layernorm_w || !L.post_attention_layernorm_w || !L.k_cache || !L.v_cache)
            return ob_fail(ONEBIT_E_ARG, "decode_step_batched: null pointer in layer %d", l);
        if ((rc = ob_check_bias_align(L, "decode_step_batched", l))) return rc;
        S3 qs = {{nullptr, nullptr, nullptr}};
        if (st->qkv_stats && NQ % 16 == 0 && NK % 16 == 0) {
            const size_t fq = (size_t)ob_tile_stats_floats(NQ), fk = (size_t)ob_tile_stats_floats(NK);
            qs.s[0] = st->qkv_stats; qs.s[1] = st->qkv_stats + (size_t)B * fq; qs.s[2] = st->qkv_stats + (size_t)B * (fq + fk);
        }
        Sk3Launch g_qkv = {}, g_o = {}, g_gu = {}, g_down = {};
        if (sk3) {
            const S3 none = {{nullptr, nullptr, nullptr}};
            g_qkv = build_sk3({&L.q, &L.k, &L.v}, {st->u_q, st->u_k, st->u_v}, qs, {xs[0], xs[1], xs[2]}, 3, H);
            g_o = o_split ? build_sk3_split2(L.o, (const _Float16 *)st->attn_out, NQ)
                          : build_sk3({&L.o, nullptr, nullptr}, {st->u_o, nullptr, nullptr}, none, {(const _Float16 *)st->attn_out, nullptr, nullptr}, 1, NQ);
            g_gu = build_sk3({&L.gate, &L.up, nullptr}, {st->u_gate, st->u_up, nullptr}, none, {xs[0], xs[1], nullptr}, 2, H);
            g_down = build_sk3_split2(L.down, (const _Float16 *)st->act, I);
        }
        // 1. residual (+ LayerNorm of the previous down_proj) + input RMSNorm
        ObBNormArgs na = {};
        na.embed = (const _Float16 *)m->embed; na.tokens = st->tokens; na.hres_in = hA;
        if (down_parts && l > 0) { na.u_prev = nullptr; na.z0 = zs0; na.z1 = zs1; na.g_prev = (const _Float16 *)m->layers[l - 1].down.weight_scale; }
        else na.u_prev = (const _Float16 *)st->u_down;
        na.rms_w = (const _Float16 *)L.input_layernorm_w; na.hres_out = hB; na.x = (_Float16 *)st->x; na.H = H;
        na.rms_eps = m->rms_eps; na.ln_eps = m->ln_eps;
        if (sk3) {
            na.x = nullptr; na.n_scaled = 3;
            na.h_next[0] = (const _Float16 *)L.q.input_factor; na.h_next[1] = (const _Float16 *)L.k.input_factor; na.h_next[2] = (const _Float16 *)L.v.input_factor;
            na.x_scaled[0] = xs[0]; na.x_scaled[1] = xs[1]; na.x_scaled[2] = xs[2];
            na.pf = plan_of(g_qkv); na.pf_rows = B;
        }
        const int pf_grid = (sk3 && sk3_pf) ? std::max(B, ob_cu_count()) : B;      // rows + one prefetch-only workgroup per idle CU
        if (l == 0) OB_LAUNCH_NORM(true, H, dim3(pf_grid), s, na);
        else OB_LAUNCH_NORM(false, H, dim3(pf_grid), s, na);
        if ((rc = ob_launch_status("decode_step_batched(norm)"))) return rc;
        // 2. q, k, v: one launch when the skinny kernel takes all three
        //    (its epilogue also publishes the LayerNorm partials of the three rows per slot, so the (head, slot)
        //    attention workgroups do not each re-reduce the whole q / k / v rows)
        if (sk3) {
            if ((rc = launch_sk3(g_qkv, "qkv"))) return rc;
            stats_written = qs.s[0] != nullptr;
        } else if ((rc = gemm_multi({&L.q, &L.k, &L.v}, {st->u_q, st->u_k, st->u_v}, {NQ, NK, NK}, qs, 3, st->x, H, "qkv"))) return rc;
        const bool attn_pst = stats_written;
        // 3. attention per (head, slot)
        ObAttnArgs at = {};
        at.u_q = (const _Float16 *)st->u_q; at.u_k = (const _Float16 *)st->u_k; at.u_v = (const _Float16 *)st->u_v;
        at.cos = (const _Float16 *)m->rope_cos; at.sin = (const _Float16 *)m->rope_sin;
        at.slot_stride = (long long)m->n_kv_heads * m->max_len * D;
        at.kcache = (_Float16 *)L.k_cache + (size_t)slot0 * at.slot_stride; at.vcache = (_Float16 *)L.v_cache + (size_t)slot0 * at.slot_stride;
        at.out = (_Float16 *)st->attn_out;
        at.pos = st->pos; at.H = m->n_heads; at.Hkv = m->n_kv_heads; at.D = D; at.max_len = m->max_len;
        at.ln_eps = m->ln_eps;
        const size_t attn_lds = 512 + 3 * 128 * 2 + (size_t)OB_ATTN_WAVES * 128 * 4 + (size_t)4 * m->max_len;
        if (st->attn_splits > 0) {
            // key-block form (ABI 9): LayerNorm + RoPE + cache append per slot, then (head, slot, split) workgroups
            const void *h_o = sk3 ? L.o.input_factor : nullptr;
            if ((L.q_bias || L.k_bias || L.v_bias) && !(L.q_bias && L.k_bias && L.v_bias))
                return ob_fail(ONEBIT_E_ARG, "decode_step_batched: layer %d has some but not all of q_bias / k_bias / v_bias", l);
            const int chunk = st->attn_chunk > 0 ? st->attn_chunk : 256;
            static const int fuse_env = getenv("OB_FDEC_FUSED") ? atoi(getenv("OB_FDEC_FUSED")) : 1;      // A/B: 0 = rope / append launch + attention
            const size_t sbytes = onebit_attention_decode_scratch_bytes(B, m->n_heads, st->attn_splits);
            if (fuse_env && attn_pst && (D & (D - 1)) == 0 && D >= 16) {
                // the q|k|v GEMM published the rows' LayerNorm partials: the attention launch forms q (k, v in the last split) itself
                if ((rc = onebit_attention_decode_rows_fused(st->u_q, st->u_k, st->u_v, qs.s[0], qs.s[1], qs.s[2], L.q_bias, L.k_bias, L.v_bias,
                                                             m->rope_cos, m->rope_sin, at.kcache, at.vcache, st->attn_out, h_o, nullptr, st->pos, B,
                                                             m->n_heads, m->n_kv_heads, D, B, m->max_len, m->max_len, chunk, st->attn_splits,
                                                             m->ln_eps, st->attn_scratch, sbytes, s)))
                    return rc;
            } else {
                if ((rc = onebit_rows_qkv_rope_ragged(st->u_q, st->u_k, st->u_v, m->rope_cos, m->rope_sin, nullptr, st->pos, st->q_rows, at.kcache, at.vcache,
                                                      L.q_bias, L.k_bias, L.v_bias, B, m->n_heads, m->n_kv_heads, D, B, m->max_len, m->max_len, m->ln_eps, s)))
                    return rc;
                if ((rc = onebit_attention_decode_rows(st->q_rows, at.kcache, at.vcache, st->attn_out, h_o, nullptr, st->pos, B, m->n_heads, m->n_kv_heads, D,
                                                       B, m->max_len, chunk, st->attn_splits, st->attn_scratch, sbytes, s)))
                    return rc;
            }
        } else {
        if (attn_lds > 64 * 1024) return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: max_len %d too large for the one-workgroup attention kernel (set attn_splits)", m->max_len);
        // 4-wave workgroups: twice as many (head, slot) workgroups resident per CU (2.95 -> 2.89 ms per 32-slot step)
        static const int battn = getenv("OB_BATCH_ATTN_THREADS") ? atoi(getenv("OB_BATCH_ATTN_THREADS")) : 256;
        if (attn_pst) { at.st_q = qs.s[0]; at.st_k = qs.s[1]; at.st_v = qs.s[2]; }
        // (the attention workgroups do not prefetch o_proj's rows: measured, the (head, slot) chains got 3.3 us longer for
        //  0.5 us off the o_proj launch; nor do the 64 CUs the q|k|v launch leaves idle: the attention launch's K / V
        //  stream runs through the same L2 in between, 2.18 vs 2.19 ms)
        if (sk3) at.h_next = (const _Float16 *)L.o.input_factor;
        at.b_q = (const _Float16 *)L.q_bias; at.b_k = (const _Float16 *)L.k_bias; at.b_v = (const _Float16 *)L.v_bias;
        const bool qkv_bias = L.q_bias || L.k_bias || L.v_bias;
        if (qkv_bias && !(L.q_bias && L.k_bias && L.v_bias))
            return ob_fail(ONEBIT_E_ARG, "decode_step_batched: layer %d has some but not all of q_bias / k_bias / v_bias", l);
        if (qkv_bias) {             // config.attention_bias: the BIAS instances
            if (attn_pst && battn == 256) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 256, false, true>), dim3(m->n_heads, B), dim3(256), attn_lds, s, at, ObPfPlan{});
            else if (attn_pst) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 512, false, true>), dim3(m->n_heads, B), dim3(512), attn_lds, s, at, ObPfPlan{});
            else if (battn == 256) hipLaunchKernelGGL((ob_dec_attn_kernel<false, 256, false, true>), dim3(m->n_heads, B), dim3(256), attn_lds, s, at, ObPfPlan{});
            else hipLaunchKernelGGL((ob_dec_attn_kernel<false, 512, false, true>), dim3(m->n_heads, B), dim3(512), attn_lds, s, at, ObPfPlan{});
        }
        else if (attn_pst && battn == 256) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 256, false>), dim3(m->n_heads, B), dim3(256), attn_lds, s, at, ObPfPlan{});
        else if (attn_pst) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 512, false>), dim3(m->n_heads, B), dim3(512), attn_lds, s, at, ObPfPlan{});
        else if (battn == 256) hipLaunchKernelGGL((ob_dec_attn_kernel<false, 256, false>), dim3(m->n_heads, B), dim3(256), attn_lds, s, at, ObPfPlan{});
        else hipLaunchKernelGGL((ob_dec_attn_kernel<false, 512, false>), dim3(m->n_heads, B), dim3(512), attn_lds, s, at, ObPfPlan{});
        if ((rc = ob_launch_status("decode_step_batched(attn)"))) return rc;
        }
        // 4. o_proj
        if (sk3) { if ((rc = launch_sk3(g_o, "o"))) return rc; }
        else if (splitk_o) { if ((rc = gemm_splitk2(L.o, st->attn_out, NQ, zs0, zs1, "o"))) return rc; }
        else if ((rc = gemm(L.o, st->attn_out, st->u_o, NQ, H, "o"))) return rc;
        // 5. residual + LayerNorm(u_o) + post-attention RMSNorm
        ObBNormArgs nb = na;
        nb.z0 = nb.z1 = nullptr; nb.g_prev = nullptr; nb.pf = {};
        nb.hres_in = hB; nb.u_prev = (const _Float16 *)st->u_o;
        if ((splitk_o && !sk3) || o_split) { nb.u_prev = nullptr; nb.z0 = zs0; nb.z1 = zs1; nb.g_prev = (const _Float16 *)L.o.weight_scale; }
        nb.rms_w = (const _Float16 *)L.post_attention_layernorm_w; nb.hres_out = hA;
        nb.bias_prev = (const _Float16 *)L.o_bias;            // o_proj's bias joins LayerNorm(u_o) here (bitnet.py:119-120)
        if (sk3) {
            nb.x = nullptr; nb.n_scaled = 2; nb.h_next[2] = nullptr; nb.x_scaled[2] = nullptr;
            nb.h_next[0] = (const _Float16 *)L.gate.input_factor; nb.h_next[1] = (const _Float16 *)L.up.input_factor;
            nb.x_scaled[0] = xs[0]; nb.x_scaled[1] = xs[1];
            nb.pf = plan_of(g_gu); nb.pf_rows = B;
        }
        OB_LAUNCH_NORM(false, H, dim3(pf_grid), s, nb);
        if ((rc = ob_launch_status("decode_step_batched(norm2)"))) return rc;
        // 6. gate, up; 7. SiLU(LN(gate)) * LN(up); 8. down
        if (sk3) {
            if ((rc = launch_sk3(g_gu, "gate|up"))) return rc;
        } else if ((rc = gemm_multi({&L.gate, &L.up, nullptr}, {st->u_gate, st->u_up, nullptr}, {I, I, 0}, {{nullptr, nullptr, nullptr}}, 2, st->x, H, "gate|up"))) return rc;
        ObBSwigluArgs sa = {(const _Float16 *)st->u_gate, (const _Float16 *)st->u_up, (_Float16 *)st->act, I, m->ln_eps,
                            sk3 ? (const _Float16 *)L.down.input_factor : nullptr, nullptr};
        if (sk3) { sa.pf = plan_of(g_down); sa.pf_rows = B; }
        OB_LAUNCH_SWIGLU(I, dim3(pf_grid), s, sa);
        if ((rc = ob_launch_status("decode_step_batched(swiglu)"))) return rc;
        // 8. down: short and wide (N = hidden, K = intermediate) -- split K over two workgroup ranges
        //    (fp32 partial sums into the free u_gate / u_up buffers), summed by the next norm kernel
        if (sk3) {
            // two K-slices of the pre-scaled SwiGLU rows, fp32 partial sums (u_gate / u_up are free: SwiGLU has consumed them)
            if ((rc = launch_sk3(g_down, "down"))) return rc;
        } else if (splitk_down) {
            if ((rc = gemm_splitk2(L.down, st->act, I, zs0, zs1, "down"))) return rc;
        } else if ((rc = gemm(L.down, st->act, st->u_down, I, H, "down"))) return rc;
    }
    // final: residual + LayerNorm(u_down) + final RMSNorm -> x
    ObBNormArgs nf = {};
    nf.hres_in = hA; nf.rms_w = (const _Float16 *)m->final_norm_w; nf.hres_out = hB;
    if (down_parts) { nf.u_prev = nullptr; nf.z0 = zs0; nf.z1 = zs1; nf.g_prev = (const _Float16 *)m->layers[m->n_layers - 1].down.weight_scale; }
    else nf.u_prev = (const _Float16 *)st->u_down;
    nf.x = (_Float16 *)st->x; nf.H = H; nf.rms_eps = m->rms_eps; nf.ln_eps = m->ln_eps;
    OB_LAUNCH_NORM(false, H, dim3(B), s, nf);
    return ob_launch_status("decode_step_batched(final norm)");
}

// side streams / events for the chain split of the batched step (per device; created on the first call, which must
// not be inside a stream capture -- callers warm up once before capturing, as every graph user of this library does)
struct ObChainCtx { hipStream_t side[3]; hipEvent_t fork, join[3]; bool ok; };
// One set of side streams / events per device, shared by every caller: creation AND use are serialised by g_chain_mu (two host
// threads, or two engines on different user streams of one device, would otherwise interleave record / wait on the same
// events).  The split is an A/B switch that measured slower than one chain; it is kept correct, not concurrent.
static std::mutex g_chain_mu;
static ObChainCtx *ob_chain_ctx()
{
    static ObChainCtx ctx[OB_MAX_DEVICES] = {};
    ObChainCtx &c = ctx[ob_device_index()];
    if (!c.ok) {
        bool good = hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; good && i < 3; ++i)
            good = hipStreamCreateWithFlags(&c.side[i], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&c.join[i], hipEventDisableTiming) == hipSuccess;
        if (!good) { (void)hipGetLastError(); return nullptr; }
        c.ok = true;
    }
    return &c;
}

extern "C" int onebit_decode_step_batched(const onebit_model_t *m, const onebit_batch_state_t *st, void *stream)
{
    if (!m || !st) return ob_fail(ONEBIT_E_ARG, "decode_step_batched: null model/state");
    if (st->struct_size != sizeof(onebit_batch_state_t))
        return ob_fail(ONEBIT_E_ARG, "decode_step_batched: state struct_size %llu != %zu (caller built against another ABI: this library is ABI %d)",
                       (unsigned long long)st->struct_size, sizeof(onebit_batch_state_t), ONEBIT_ABI_VERSION);
    if (m->n_layers <= 0 || m->hidden <= 0 || m->n_heads <= 0 || m->n_kv_heads <= 0 || m->head_dim <= 0 ||
        m->n_heads % m->n_kv_heads != 0 || m->head_dim % 8 != 0 || m->head_dim > 128 || m->hidden % 8 != 0 ||
        m->intermediate % 8 != 0 || m->max_len <= 0)
        return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: bad model dimensions");
    if (st->batch < 2 || st->batch > 64) return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: batch %d outside 2..64", st->batch);
    if (m->hidden > OB_DEC_MAXV * OB_DEC_THREADS * 8 || m->intermediate > OB_DEC_MAXV * OB_DEC_THREADS * 8)
        return ob_fail(ONEBIT_E_SHAPE, "decode_step_batched: hidden / intermediate beyond %d", OB_DEC_MAXV * OB_DEC_THREADS * 8);
    if (!m->layers || !m->embed || !m->final_norm_w || !m->rope_cos || !m->rope_sin || !st->tokens || !st->pos ||
        !st->hres0 || !st->hres1 || !st->x || !st->act || !st->u_q || !st->u_k || !st->u_v || !st->attn_out ||
        !st->u_o || !st->u_gate || !st->u_up || !st->u_down)
        return ob_fail(ONEBIT_E_ARG, "decode_step_batched: null pointer");
    if (st->attn_splits < 0 || (st->attn_splits > 0 && (!st->q_rows || (st->attn_splits > 1 && !st->attn_scratch) ||
                                                         (st->attn_chunk != 0 && (st->attn_chunk < 64 || st->attn_chunk % 64 != 0)))))
        return ob_fail(ONEBIT_E_ARG, "decode_step_batched: attn_splits needs q_rows, attn_scratch (splits > 1) and attn_chunk a multiple of 64");
    hipStream_t s = (hipStream_t)stream;
    const int B = st->batch, H = m->hidden, I = m->intermediate;
    const int NQ = m->n_heads * m->head_dim, NK = m->n_kv_heads * m->head_dim;
    // Independent chains: the B sequences as `chains` groups of consecutive slots, each group's layer chain on its own
    // stream (forked from / joined to `stream`; under capture: parallel graph branches), ONE lm_head over all rows at the
    // end.  A chain is 8 dependent launches per layer on an otherwise idle chip -- one chain's GEMM fills the other's
    // ramps, tails, row kernels and attention; the second read of a layer's packed rows comes from L2 / Infinity Cache.
    // Per-row results do not depend on the grouping (a row's sums never mix with other rows').
    static const int chains_env = getenv("OB_BATCH_CHAINS") ? atoi(getenv("OB_BATCH_CHAINS")) : 0;
    int chains = chains_env > 0 ? chains_env : (st->chains > 0 ? st->chains : 1);
    if (chains > 4) chains = 4;
    while (chains > 1 && B / chains < 2) --chains;
    if (st->attn_splits > 0) chains = 1;             // (the split attention's scratch is laid out for all B rows of one chain)
    int rc;
    if (chains <= 1) {
        if ((rc = ob_batched_layers(m, st, 0, s))) return rc;
    } else {
        std::lock_guard<std::mutex> chain_lock(g_chain_mu);
        ObChainCtx *cx = ob_chain_ctx();
        if (!cx) return ob_fail(ONEBIT_E_ARG, "decode_step_batched: cannot create the side streams of the chain split");
        const size_t fq = (size_t)ob_tile_stats_floats(NQ), fk = (size_t)ob_tile_stats_floats(NK);
        if (hipEventRecord(cx->fork, s) != hipSuccess) return ob_launch_status("decode_step_batched(fork)");
        const int per = B / chains;
        int forked = 0;                 // side streams that wait on the fork: every one of them is joined, also on an error return
        auto join_all = [&]() -> bool { // (an unjoined side stream would invalidate an active stream capture)
            bool ok = true;
            for (int c = 1; c <= forked; ++c)
                ok = hipEventRecord(cx->join[c - 1], cx->side[c - 1]) == hipSuccess && hipStreamWaitEvent(s, cx->join[c - 1], 0) == hipSuccess && ok;
            return ok;
        };
        for (int c = 0; c < chains; ++c) {
            const int r0 = c * per, nb = c == chains - 1 ? B - r0 : per;
            onebit_batch_state_t v = *st;
            v.batch = nb;
            v.tokens = st->tokens + r0; v.pos = st->pos + r0;
            auto row = [&](void *p, size_t width) -> void * { return p ? (void *)((_Float16 *)p + (size_t)r0 * width) : nullptr; };
            v.hres0 = row(st->hres0, H); v.hres1 = row(st->hres1, H); v.x = row(st->x, H); v.act = row(st->act, I);
            v.u_q = row(st->u_q, NQ); v.u_k = row(st->u_k, NK); v.u_v = row(st->u_v, NK); v.attn_out = row(st->attn_out, NQ);
            v.u_o = row(st->u_o, H); v.u_gate = row(st->u_gate, I); v.u_up = row(st->u_up, I); v.u_down = row(st->u_down, H);
            v.qkv_stats = st->qkv_stats ? st->qkv_stats + (size_t)r0 * (fq + 2 * fk) : nullptr;     // a group's [3][nb] blocks
            v.x_scaled = st->x_scaled ? (void *)((_Float16 *)st->x_scaled + (size_t)r0 * 3 * H) : nullptr;   // a group's [3][nb][H]
            hipStream_t cs = c == 0 ? s : cx->side[c - 1];
            if (c > 0) {
                if (hipStreamWaitEvent(cs, cx->fork, 0) != hipSuccess) { (void)join_all(); return ob_launch_status("decode_step_batched(fork wait)"); }
                forked = c;
            }
            if ((rc = ob_batched_layers(m, &v, r0, cs))) { (void)join_all(); return rc; }
        }
        if (!join_all()) return ob_launch_status("decode_step_batched(join)");
    }
    return ob_batched_head(m, st, s);
}

int ob_lm_head_argmax(const void *x, const void *lm_head, void *logits, float *part_val, int32_t *part_idx, int32_t *next_tokens,
                      int B, int H, int V, hipStream_t s)
{
    int rc;
    if (B < 1 || B > 64) return ob_fail(ONEBIT_E_SHAPE, "lm_head: %d rows outside 1..64", B);
    if (H % 64 != 0) return ob_fail(ONEBIT_E_SHAPE, "lm_head needs hidden %% 64 == 0");
    ObBHeadArgs ha = {(const _Float16 *)x, (const _Float16 *)lm_head, (_Float16 *)logits, part_val, part_idx, B, H, V};
    const int hg = (V + OB_BH_ROWS - 1) / OB_BH_ROWS;
#define OB_BH_GO(TT_)                                                                                              \
    do {                                                                                                           \
        const size_t lds = (size_t)2 * TT_ * 16 * OB_BH_PITCH * 2 + (size_t)8 * TT_ * 16 * 8;                      \
        static bool attr_set[OB_MAX_DEVICES] = {};                                                                 \
        ob_set_max_lds_once(ob_b_lmhead_kernel<TT_>, attr_set, (int)lds);                                          \
        hipLaunchKernelGGL(ob_b_lmhead_kernel<TT_>, dim3(hg), dim3(512), lds, s, ha);                              \
    } while (0)
    if (B <= 16) OB_BH_GO(1);
    else if (B <= 32) OB_BH_GO(2);
    else OB_BH_GO(4);
#undef OB_BH_GO
    if ((rc = ob_launch_status("lm_head"))) return rc;
    hipLaunchKernelGGL(ob_b_argmax_kernel, dim3(B), dim3(256), 0, s, (const float *)part_val, (const int *)part_idx, hg, V, next_tokens);
    return ob_launch_status("lm_head(argmax)");
}

static int ob_batched_head(const onebit_model_t *m, const onebit_batch_state_t *st, hipStream_t s)
{
    if (!st->next_tokens) return 0;
    // batched lm_head + greedy sampling
    if (!m->lm_head || m->vocab <= 0 || !st->part_val || !st->part_idx)
        return ob_fail(ONEBIT_E_ARG, "decode_step_batched: next_tokens needs model->lm_head, vocab and the part_val / part_idx scratch");
    return ob_lm_head_argmax(st->x, m->lm_head, st->logits, st->part_val, st->part_idx, st->next_tokens, st->batch, m->hidden, m->vocab, s);
}

// ---- launchers the mixed step (onebit_mixed.hip) drives: ob_host.h ----
int ob_rows_norm(const ObRowsNormCall &c, hipStream_t s)
{
    if (c.T <= 0) return 0;
    if (c.H <= 0 || c.H % 8 != 0 || c.H > OB_DEC_MAXV * OB_DEC_THREADS * 8 || c.n_scaled < 0 || c.n_scaled > 3 || c.T > 0x7fffffffLL)
        return ob_fail(ONEBIT_E_SHAPE, "rows_norm: bad shape");
    if (!c.rms_w || !c.hres_out || (!c.x && c.n_scaled == 0) || (c.embed ? !c.tokens : (!c.hres_in || (!c.u_prev && !(c.z0 && c.g_prev)))))
        return ob_fail(ONEBIT_E_ARG, "rows_norm: null pointer");
    ObBNormArgs a = {};
    a.embed = (const _Float16 *)c.embed; a.tokens = c.tokens; a.hres_in = (const _Float16 *)c.hres_in; a.u_prev = (const _Float16 *)c.u_prev;
    a.bias_prev = (const _Float16 *)c.bias_prev; a.rms_w = (const _Float16 *)c.rms_w; a.hres_out = (_Float16 *)c.hres_out; a.x = (_Float16 *)c.x;
    a.H = c.H; a.rms_eps = c.rms_eps; a.ln_eps = c.ln_eps; a.n_scaled = c.n_scaled; a.rows = c.rows;
    if (!c.embed && !c.u_prev) { a.z0 = c.z0; a.z1 = c.z1; a.z2 = c.z2; a.z3 = c.z3; a.g_prev = (const _Float16 *)c.g_prev; }
    for (int i = 0; i < c.n_scaled; ++i) {
        if (!c.h_next[i] || !c.x_scaled[i]) return ob_fail(ONEBIT_E_ARG, "rows_norm: null scaled output %d", i);
        a.h_next[i] = (const _Float16 *)c.h_next[i]; a.x_scaled[i] = (_Float16 *)c.x_scaled[i];
    }
    if (c.embed) OB_LAUNCH_NORM(true, c.H, dim3((unsigned)c.T), s, a);
    else OB_LAUNCH_NORM(false, c.H, dim3((unsigned)c.T), s, a);
    return ob_launch_status("rows_norm");
}

bool ob_gemm3_group_ok(const onebit_proj_t *const *ps, int np, int64_t T)
{
    static const int env = getenv("OB_GEMM3_GROUPED") ? atoi(getenv("OB_GEMM3_GROUPED")) : 1;
    static const int env3 = getenv("OB_GEMM3") ? atoi(getenv("OB_GEMM3")) : 1;
    if (!env || !env3 || np < 2 || np > 3) return false;
    const int64_t K = ps[0]->K, ldw = ps[0]->ldw_bytes;
    if (T < 192 || K % (4 * OB_G2_K) != 0 || T * K * 2 >= ((int64_t)1 << 32) || ldw % 16 != 0) return false;
    int64_t tiles = 0;
    for (int i = 0; i < np; ++i) {
        const onebit_proj_t &p = *ps[i];
        if (!p.weight || !p.weight_scale || p.K != K || p.ldw_bytes != ldw || p.N % 4 != 0 || p.N * (K / 8) >= ((int64_t)1 << 32) ||
            p.N * ldw >= ((int64_t)1 << 32) || !ob_aligned(p.weight, 16) || !ob_aligned(p.weight_scale, 8))
            return false;
        tiles += ((p.N + OB_G2_N - 1) / OB_G2_N) * ((T + 127) / 128);
    }
    // a GROUP is worth one launch from half a round of workgroups on: the alternative is one launch of the round-1 kernel per member
    static const int grp_den = getenv("OB_GEMM3_GROUP_DEN") ? atoi(getenv("OB_GEMM3_GROUP_DEN")) : 2;
    return tiles * std::max(grp_den, 1) >= (int64_t)ob_cu_count() && tiles <= 0x3fffffff;
}

int64_t ob_gemm3_group_tiles(const onebit_proj_t *const *ps, int np, int64_t T)
{
    int64_t tiles = 0;
    for (int i = 0; i < np; ++i) tiles += ((ps[i]->N + OB_G2_N - 1) / OB_G2_N) * ((T + 127) / 128);
    return tiles;
}
int ob_gemm3_slots() { return 2 * ob_cu_count(); }

int ob_gemm3_grouped(const onebit_proj_t *const *ps, void *const *us, const void *const *as, int np, int64_t T, hipStream_t s)
{
    static const int env = getenv("OB_GEMM3_GROUPED") ? atoi(getenv("OB_GEMM3_GROUPED")) : 1;       // A/B: 0 = one launch per projection
    if (!env || np < 2 || np > 3) return ob_fail(ONEBIT_E_SHAPE, "gemm3_grouped: 2..3 projections");
    const int64_t K = ps[0]->K, ldw = ps[0]->ldw_bytes;
    if (T < 192 || K % (4 * OB_G2_K) != 0 || T * K * 2 >= ((int64_t)1 << 32) || ldw % 16 != 0)
        return ob_fail(ONEBIT_E_SHAPE, "gemm3_grouped: shape not eligible");
    ObG3Group G = {};
    const int nbt = (int)((T + 127) / 128);
    int64_t tiles = 0;
    for (int i = 0; i < 3; ++i) {
        const int j = i < np ? i : np - 1;
        const onebit_proj_t &p = *ps[j];
        if (i < np) {
            if (!p.weight || !p.weight_scale || !us[i] || !as[i] || p.K != K || p.ldw_bytes != ldw || p.N % 4 != 0 || p.N * (K / 8) >= ((int64_t)1 << 32) ||
                p.N * ldw >= ((int64_t)1 << 32) || !ob_aligned(p.weight, 16) || !ob_aligned(as[i], 16) || !ob_aligned(us[i], 16) || !ob_aligned(p.weight_scale, 8))
                return ob_fail(ONEBIT_E_SHAPE, "gemm3_grouped: projection %d not eligible", i);
            G.nbn[i] = (int)((p.N + OB_G2_N - 1) / OB_G2_N);
            tiles += (int64_t)G.nbn[i] * nbt;
        } else G.nbn[i] = G.nbn[j];
        G.W[i] = (const uint32_t *)p.weight; G.a[i] = (const _Float16 *)as[j]; G.g[i] = (const _Float16 *)p.weight_scale; G.u[i] = (_Float16 *)us[j];
        G.N[i] = (int)p.N; G.tile_end[i] = (int)tiles;
    }
    if (!ob_gemm3_group_ok(ps, np, T)) return ob_fail(ONEBIT_E_SHAPE, "gemm3_grouped: grid too small / large");
    G.ldw_words = ldw / 4; G.lda = K; G.T = (int)T; G.K = (int)K;
    static const int g4_env = getenv("OB_GEMM4") ? atoi(getenv("OB_GEMM4")) : OB_GEMM4_DEFAULT;
    if (g4_env) {
        static bool attr_set4[OB_MAX_DEVICES] = {};
        ob_set_max_lds_once(ob_gemm4g_f16_kernel, attr_set4, OB_G4_LDS);
        hipLaunchKernelGGL(ob_gemm4g_f16_kernel, dim3((unsigned)tiles), dim3(256), OB_G4_LDS, s, G);
        return ob_launch_status("gemm4_grouped");
    }
    static bool attr_set[OB_MAX_DEVICES] = {};
    ob_set_max_lds_once(ob_gemm3g_f16_kernel<1>, attr_set, OB_G3_LDS_W(1));
    hipLaunchKernelGGL((ob_gemm3g_f16_kernel<1>), dim3((unsigned)tiles), dim3(256), OB_G3_LDS_W(1), s, G);
    return ob_launch_status("gemm3_grouped");
}

int ob_gemm3_ksplit_n(const onebit_proj_t &p, int64_t T)
{
    static const int env = getenv("OB_GEMM3_KSPLIT") ? atoi(getenv("OB_GEMM3_KSPLIT")) : 4;         // A/B: 0 = off, 2 .. 4 = that many slices at most
    static const int env3 = getenv("OB_GEMM3") ? atoi(getenv("OB_GEMM3")) : 1;
    // a projection with fewer tiles than this is sliced (default: what the unsliced LDS-DMA GEMM asks for, two thirds of the CUs)
    static const int tiles_env = getenv("OB_GEMM3_KSPLIT_TILES") ? atoi(getenv("OB_GEMM3_KSPLIT_TILES")) : 0;
    static const int tmin = getenv("OB_GEMM3_KSPLIT_TMIN") ? atoi(getenv("OB_GEMM3_KSPLIT_TMIN")) : 65;
    if (env < 2 || !env3 || T < tmin || p.K % (4 * OB_G2_K) != 0 || p.N % 4 != 0 || T * p.K * 2 >= ((int64_t)1 << 32) || p.ldw_bytes % 16 != 0 ||
        p.N * p.ldw_bytes >= ((int64_t)1 << 32) || !p.weight || !p.weight_scale || !ob_aligned(p.weight, 16) || !ob_aligned(p.weight_scale, 16))
        return 0;
    const int64_t tiles = ((p.N + OB_G2_N - 1) / OB_G2_N) * ((T + 127) / 128);
    const int64_t cus = ob_cu_count();
    if (tiles_env ? tiles >= tiles_env : 3 * tiles >= 2 * cus) return 0;      // enough tiles alone
    // as many slices as leave each at least four quads (1024 columns) of K loop: the shorter the loop, the shorter the one round
    const int quads = (int)(p.K / (4 * OB_G2_K));
    const int ns = std::min(std::min(env, 4), quads / 4);
    return ns >= 2 ? ns : 0;
}

int ob_gemm3_ksplit(const onebit_proj_t &p, const void *a, float *const *z, int ns, int64_t T, hipStream_t s)
{
    if (ns < 2 || ns > 4 || ns != ob_gemm3_ksplit_n(p, T) || !a || !z || !ob_aligned(a, 16))
        return ob_fail(ONEBIT_E_SHAPE, "gemm3_ksplit: not eligible");
    ObG3Slices G = {};
    for (int i = 0; i < 4; ++i) {
        G.z[i] = z[i < ns ? i : ns - 1];
        if (!G.z[i] || !ob_aligned(G.z[i], 16)) return ob_fail(ONEBIT_E_ALIGN, "gemm3_ksplit: partial sums must be 16-byte aligned");
    }
    G.W = (const uint32_t *)p.weight; G.a = (const _Float16 *)a; G.ldw_words = p.ldw_bytes / 4; G.lda = p.K; G.T = (int)T; G.N = (int)p.N;
    G.nbn = (int)((p.N + OB_G2_N - 1) / OB_G2_N); G.tiles = G.nbn * (int)((T + 127) / 128); G.ns = ns; G.quads = (int)(p.K / (4 * OB_G2_K));
    static bool attr_set[OB_MAX_DEVICES] = {};
    ob_set_max_lds_once(ob_gemm3ks_f16_kernel<1>, attr_set, OB_G3_LDS_W(1));
    hipLaunchKernelGGL((ob_gemm3ks_f16_kernel<1>), dim3((unsigned)(ns * G.tiles)), dim3(256), OB_G3_LDS_W(1), s, G);
    return ob_launch_status("gemm3_ksplit");
}

bool ob_sk3_proj_ok(const onebit_proj_t &p)
{
    return p.weight && p.weight_scale && ob_aligned(p.weight_scale, 16) && ob_skinny3_shape_ok(p.weight, p.ldw_bytes, nullptr, p.K, 64, p.K, p.N);
}

int ob_sk3_multi(const onebit_proj_t *const *ps, void *const *us, const void *const *as, int np, int64_t T, hipStream_t s)
{
    if (np < 1 || np > 3) return ob_fail(ONEBIT_E_ARG, "sk3_multi: 1..3 projections");
    const int64_t K = ps[0]->K;
    int64_t nn[3] = {0, 0, 0};
    for (int i = 0; i < np; ++i) {
        const onebit_proj_t &p = *ps[i];
        if (!p.weight || !p.weight_scale || p.K != K || !us[i] || !as[i] || !ob_skinny3_shape_ok(p.weight, p.ldw_bytes, as[i], K, T, K, p.N))
            return ob_fail(ONEBIT_E_SHAPE, "sk3_multi: projection %d is not eligible for the LDS-DMA skinny GEMM", i);
        nn[i] = p.N;
    }
    const int rnt = ob_skinny3_pick_rnt(nn, np, 1);
    ObSk3Args ka = {};
    int wgs = 0;
    for (int i = 0; i < 3; ++i) {
        const int j = i < np ? i : np - 1;
        const onebit_proj_t &p = *ps[j];
        if (i < np) wgs += (int)((p.N + 16 * rnt - 1) / (16 * rnt));
        ka.p[i] = {(const uint32_t *)p.weight, (long long)(p.ldw_bytes / 4), (const _Float16 *)p.weight_scale, (const _Float16 *)as[j],
                   (_Float16 *)us[j], nullptr, nullptr, (int)p.N, (int)K, wgs};
    }
    ka.lda = K; ka.T = (int)T;
    if (!ob_launch_skinny3<false>(ka, wgs, rnt, s)) return ob_fail(ONEBIT_E_SHAPE, "sk3_multi: no skinny GEMM instance");
    return ob_launch_status("sk3_multi");
}

extern "C" int onebit_fused_gemv(const onebit_proj_t *projs, void *const *outs, int nproj, int prologue,
                                 const onebit_fused_in_t *in, void *stream)
{
    if (!projs || !outs || !in || nproj < 1 || nproj > 3) return ob_fail(ONEBIT_E_ARG, "fused_gemv: bad arguments");
    ObGemvArgs a = {};
    a.nproj = nproj; a.K = (int)projs[0].K; a.prologue = prologue;
    int rc;
    for (int p = 0; p < nproj; ++p) {
        if (projs[p].K != projs[0].K) return ob_fail(ONEBIT_E_SHAPE, "fused_gemv: projections must share in_features");
        if ((rc = ob_fill_proj(a.p[p], projs[p], outs[p], "fused", in->st_out[p]))) return rc;
    }
    a.st_prev = in->st_prev; a.st_gate = in->st_gate; a.st_up = in->st_up;
    if (a.K % 16 != 0 && (a.st_prev || a.st_gate || a.st_up))
        return ob_fail(ONEBIT_E_SHAPE, "fused_gemv: tile statistics need in_features %% 16 == 0");
    for (const float *sp : {a.st_prev, a.st_gate, a.st_up, (const float *)in->st_out[0], (const float *)in->st_out[1], (const float *)in->st_out[2]})
        if (sp && !ob_aligned(sp, 16)) return ob_fail(ONEBIT_E_ALIGN, "fused_gemv: tile statistics must be 16-byte aligned");
    a.xin = (const _Float16 *)in->xin; a.embed = (const _Float16 *)in->embed; a.token = in->token;
    a.hres_in = (const _Float16 *)in->hres_in; a.u_prev = (const _Float16 *)in->u_prev;
    a.hres_out = (_Float16 *)in->hres_out; a.rms_w = (const _Float16 *)in->rms_w;
    a.u_gate = (const _Float16 *)in->u_gate; a.u_up = (const _Float16 *)in->u_up;
    a.rms_eps = in->rms_eps; a.ln_eps = in->ln_eps;
    bool ok = false;
    switch (prologue) {
    case OB_P_PLAIN: ok = a.xin != nullptr; break;
    case OB_P_EMBED_RMS: ok = a.embed && a.token && a.rms_w; break;
    case OB_P_RES_LN_RMS: ok = a.hres_in && a.u_prev && a.rms_w; break;
    case OB_P_SWIGLU: ok = a.u_gate && a.u_up; break;
    default: return ob_fail(ONEBIT_E_FLAG, "fused_gemv: unknown prologue %d", prologue);
    }
    if (!ok) return ob_fail(ONEBIT_E_ARG, "fused_gemv: null input for prologue %d", prologue);
    return ob_launch_dec_gemv(a, (hipStream_t)stream);
}

extern "C" int onebit_decode_step(const onebit_model_t *m, const onebit_decode_state_t *st, void *stream)
{
    if (!m || !st || !m->layers) return ob_fail(ONEBIT_E_ARG, "decode_step: null model/state");
    if (st->struct_size != sizeof(onebit_decode_state_t))
        return ob_fail(ONEBIT_E_ARG, "decode_step: state struct_size %llu != %zu (caller built against another ABI: this library is ABI %d)",
                       (unsigned long long)st->struct_size, sizeof(onebit_decode_state_t), ONEBIT_ABI_VERSION);
    if (m->n_layers <= 0 || m->hidden <= 0 || m->n_heads <= 0 || m->n_kv_heads <= 0 || m->head_dim <= 0 ||
        m->n_heads % m->n_kv_heads != 0 || m->head_dim % 8 != 0 || m->head_dim > 128 || m->hidden % 8 != 0 ||
        m->intermediate % 8 != 0 || m->max_len <= 0 || m->vocab <= 0)
        return ob_fail(ONEBIT_E_SHAPE, "decode_step: unsupported model dimensions");
    if (!st->token || !st->pos || !st->hres0 || !st->hres1 || !st->u_q || !st->u_k || !st->u_v || !st->attn_out ||
        !st->u_o || !st->u_gate || !st->u_up || !st->u_down || !st->logits || !st->part_val || !st->part_idx ||
        !m->embed || !m->final_norm_w || !m->lm_head || !m->rope_cos || !m->rope_sin)
        return ob_fail(ONEBIT_E_ARG, "decode_step: null buffer");
    // tile_stats == NULL: every consumer recomputes its LayerNorm statistics from the vectors -- the kernels' non-PST forms
    if (st->tile_stats && !ob_aligned(st->tile_stats, 16))
        return ob_fail(ONEBIT_E_ALIGN, "decode_step: tile_stats must be 16-byte aligned");
    // (every flag is validated HERE, before the first launch: a bad value must not leave a half-issued step or a half-built capture)
    if (st->attn_blind != 0 && st->attn_blind != 64 && st->attn_blind != 128)
        return ob_fail(ONEBIT_E_FLAG, "decode_step: attn_blind %d (0, 64 or 128)", st->attn_blind);
    static const int blind_env = getenv("OB_ATTN_BLIND") ? atoi(getenv("OB_ATTN_BLIND")) : 0;      // A/B override: 64 / 128
    if (blind_env != 0 && blind_env != 64 && blind_env != 128)
        return ob_fail(ONEBIT_E_FLAG, "decode_step: OB_ATTN_BLIND=%d (64 or 128)", blind_env);
    const bool keyblock = st->attn_chunk > 0;
    if (keyblock && (st->attn_chunk % 64 != 0 || st->attn_splits < 1 || st->attn_splits > 64 || !st->q_rows || (st->attn_splits > 1 && !st->attn_scratch)))
        return ob_fail(ONEBIT_E_ARG, "decode_step: attn_chunk %d needs a multiple of 64, 1..64 attn_splits, q_rows and (splits > 1) attn_scratch", st->attn_chunk);
    hipStream_t s = (hipStream_t)stream;
    const int H = m->hidden, I = m->intermediate, D = m->head_dim;
    _Float16 *hA = (_Float16 *)st->hres0, *hB = (_Float16 *)st->hres1;
    // tile partials of the seven pre-LayerNorm vectors of a layer (reused by every layer)
    ObStatsLayout sl = ob_stats_layout(m);
    float *ts = st->tile_stats;
    float *ts_q = ts + sl.off[0], *ts_k = ts + sl.off[1], *ts_v = ts + sl.off[2], *ts_o = ts + sl.off[3],
          *ts_gate = ts + sl.off[4], *ts_up = ts + sl.off[5], *ts_down = ts + sl.off[6];
    if (!ts || (m->n_heads * D) % 16 || (m->n_kv_heads * D) % 16 || H % 16 || I % 16)  // no buffer / partial tiles: every consumer
        ts_q = ts_k = ts_v = ts_o = ts_gate = ts_up = ts_down = nullptr;             // recomputes its statistics
    int rc;
    for (int l = 0; l < m->n_layers; ++l) {
        const onebit_layer_t &L = m->layers[l];
        if (!L.k_cache || !L.v_cache || !L.input_layernorm_w || !L.post_attention_layernorm_w)
            return ob_fail(ONEBIT_E_ARG, "decode_step: null buffer in layer %d", l);
        const bool qkv_bias = L.q_bias || L.k_bias || L.v_bias;
        if (qkv_bias && !(L.q_bias && L.k_bias && L.v_bias))
            return ob_fail(ONEBIT_E_ARG, "decode_step: layer %d has some but not all of q_bias / k_bias / v_bias", l);
        if ((rc = ob_check_bias_align(L, "decode_step", l))) return rc;
        // K1: residual (+LN of previous down) -> RMSNorm -> q, k, v
        ObGemvArgs a = {};
        a.nproj = 3; a.K = H;
        if ((rc = ob_fill_proj(a.p[0], L.q, st->u_q, "q_proj", ts_q))) return rc;
        if ((rc = ob_fill_proj(a.p[1], L.k, st->u_k, "k_proj", ts_k))) return rc;
        if ((rc = ob_fill_proj(a.p[2], L.v, st->u_v, "v_proj", ts_v))) return rc;
        if (L.q.K != H || L.k.K != H || L.v.K != H || L.q.N != (int64_t)m->n_heads * D || L.k.N != (int64_t)m->n_kv_heads * D ||
            L.v.N != L.k.N || L.o.K != L.q.N || L.o.N != H || L.gate.K != H || L.up.K != H || L.gate.N != I ||
            L.up.N != I || L.down.K != I || L.down.N != H)
            return ob_fail(ONEBIT_E_SHAPE, "decode_step: layer %d projection shapes do not match the model", l);
        a.prologue = l == 0 ? OB_P_EMBED_RMS : OB_P_RES_LN_RMS;
        a.embed = (const _Float16 *)m->embed; a.token = st->token;
        a.hres_in = hA; a.u_prev = (const _Float16 *)st->u_down; a.hres_out = hB;
        a.st_prev = ts_down;
        a.rms_w = (const _Float16 *)L.input_layernorm_w;
        a.rms_eps = m->rms_eps; a.ln_eps = m->ln_eps;
        // single-launch attention only (the split-KV kernels read the position first anyway)
        const bool rope_cur = st->rope_cur && !(st->attn_splits > 1 && st->attn_scratch) && !keyblock;
        if (l == 0 && rope_cur) {
            a.rope_pos = st->pos; a.rope_cos = (const _Float16 *)m->rope_cos; a.rope_sin = (const _Float16 *)m->rope_sin;
            a.rope_out = (_Float16 *)st->rope_cur; a.rope_D = D; a.rope_max = m->max_len;
        }
        if ((rc = ob_launch_dec_gemv(a, s))) return rc;
        // K3's launch description first: the attention launch's idle CUs prefetch its packed rows
        ObGemvArgs o = {};
        o.nproj = 1; o.K = (int)L.o.K; o.prologue = OB_P_PLAIN; o.xin = (const _Float16 *)st->attn_out;
        if ((rc = ob_fill_proj(o.p[0], L.o, st->u_o, "o_proj", ts_o))) return rc;
        o.rms_eps = m->rms_eps; o.ln_eps = m->ln_eps;
        // K2: attention
        ObAttnArgs at = {};
        at.u_q = (const _Float16 *)st->u_q; at.u_k = (const _Float16 *)st->u_k; at.u_v = (const _Float16 *)st->u_v;
        at.cos = (const _Float16 *)m->rope_cos; at.sin = (const _Float16 *)m->rope_sin;
        at.kcache = (_Float16 *)L.k_cache; at.vcache = (_Float16 *)L.v_cache; at.out = (_Float16 *)st->attn_out;
        at.pos = st->pos; at.H = m->n_heads; at.Hkv = m->n_kv_heads; at.D = D; at.max_len = m->max_len;
        at.ln_eps = m->ln_eps;
        at.st_q = ts_q; at.st_k = ts_k; at.st_v = ts_v;
        if (rope_cur) at.rope_cur = (const _Float16 *)st->rope_cur;
        at.b_q = (const _Float16 *)L.q_bias; at.b_k = (const _Float16 *)L.k_bias; at.b_v = (const _Float16 *)L.v_bias;
        if (keyblock) {
            static const int fuse_env = getenv("OB_FDEC_FUSED") ? atoi(getenv("OB_FDEC_FUSED")) : 1;      // A/B: 0 = rope / append launch + attention
            const size_t sbytes = onebit_attention_decode_scratch_bytes(1, m->n_heads, st->attn_splits);
            if (fuse_env && ts_q && (D & (D - 1)) == 0 && D >= 16) {
                // the q|k|v GEMV published the vectors' LayerNorm partials: the attention launch forms q (k, v in the last split) itself
                if ((rc = onebit_attention_decode_rows_fused(st->u_q, st->u_k, st->u_v, ts_q, ts_k, ts_v, L.q_bias, L.k_bias, L.v_bias, m->rope_cos,
                                                             m->rope_sin, L.k_cache, L.v_cache, st->attn_out, nullptr, nullptr, st->pos, 1, m->n_heads,
                                                             m->n_kv_heads, D, 1, m->max_len, m->max_len, st->attn_chunk, st->attn_splits, m->ln_eps,
                                                             st->attn_scratch, sbytes, s)))
                    return rc;
            } else {
                if ((rc = onebit_rows_qkv_rope_ragged(st->u_q, st->u_k, st->u_v, m->rope_cos, m->rope_sin, nullptr, st->pos, st->q_rows, L.k_cache, L.v_cache,
                                                      L.q_bias, L.k_bias, L.v_bias, 1, m->n_heads, m->n_kv_heads, D, 1, m->max_len, m->max_len, m->ln_eps, s)))
                    return rc;
                if ((rc = onebit_attention_decode_rows(st->q_rows, L.k_cache, L.v_cache, st->attn_out, nullptr, nullptr, st->pos, 1, m->n_heads, m->n_kv_heads, D,
                                                       1, m->max_len, st->attn_chunk, st->attn_splits, st->attn_scratch, sbytes, s)))
                    return rc;
            }
        } else if (st->attn_splits > 1 && st->attn_scratch) {
            const int S = st->attn_splits;
            if (S > 16) return ob_fail(ONEBIT_E_SHAPE, "decode_step: attn_splits %d > 16", S);
            ObAttnSplitArgs sp = {};
            sp.a = at; sp.S = S; sp.chunk = ((m->max_len + S - 1) / S + 31) & ~31;
            char *base = (char *)st->attn_scratch;
            sp.scores = (float *)base;                      base += (size_t)m->n_heads * m->max_len * 4;
            sp.stats = (float *)base;                       base += (size_t)m->n_heads * S * 2 * 4;
            sp.part = (float *)base;                        base += (size_t)m->n_heads * S * 128 * 4;
            sp.counter = (int *)base;
            const size_t lds_a = 512 + 2 * 128 * 2 + (size_t)sp.chunk * 4, lds_b = (size_t)OB_ATTN_WAVES * 128 * 4 + 16;
            if (lds_a > 64 * 1024) return ob_fail(ONEBIT_E_SHAPE, "decode_step: max_len %d too large for %d attention splits", m->max_len, S);
            hipLaunchKernelGGL(ob_dec_attn_scores_kernel, dim3(m->n_heads, S), dim3(OB_ATTN_THREADS), lds_a, s, sp);
            if ((rc = ob_launch_status("decode_step(attn scores)"))) return rc;
            hipLaunchKernelGGL(ob_dec_attn_pv_kernel, dim3(m->n_heads, S), dim3(OB_ATTN_THREADS), lds_b, s, sp);
            if ((rc = ob_launch_status("decode_step(attn pv)"))) return rc;
        } else {
            const size_t attn_lds = 512 + 3 * 128 * 2 + (size_t)OB_ATTN_WAVES * 128 * 4 + (size_t)4 * m->max_len;
            if (attn_lds > 64 * 1024) return ob_fail(ONEBIT_E_SHAPE, "decode_step: max_len %d too large for the attention kernel", m->max_len);
            // one wave per SIMD (256 threads) for the single sequence; OB_ATTN_THREADS=512 restores 8 waves (A/B)
            static const int attn_threads = getenv("OB_ATTN_THREADS") ? atoi(getenv("OB_ATTN_THREADS")) : 256;
            // o_proj's packed rows (the next launch, 2 MB at 7B) are pulled into L2 by one extra workgroup per CU the heads
            // leave idle: 986-989 -> 1004 tok/s on one box, alternating runs.  (The same for gate|up's 11 MB, or from the
            // GEMV launches' tails, loses: DESIGN.md section 6.)
            static const int attn_pf_env = getenv("OB_DEC_PREFETCH_O") ? atoi(getenv("OB_DEC_PREFETCH_O")) : 1;
            const bool blind64 = (blind_env ? blind_env : st->attn_blind) == 64;
            ObPfPlan apf = {};
            if (attn_pf_env && m->n_heads < ob_cu_count()) apf = ob_dec_gemv_plan(o);
            const int agrid = apf.nseg ? ob_cu_count() : m->n_heads;
            if (qkv_bias) {         // config.attention_bias: the BIAS instances (q / k / v = fp16(LayerNorm(u) + b) before RoPE)
                if (at.st_q && attn_threads == 256 && blind64) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 256, true, true, false, 64>), dim3(agrid), dim3(256), attn_lds, s, at, apf);
                else if (at.st_q && attn_threads == 256) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 256, true, true>), dim3(agrid), dim3(256), attn_lds, s, at, apf);
                else if (at.st_q) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 512, true, true>), dim3(agrid), dim3(512), attn_lds, s, at, apf);
                else hipLaunchKernelGGL((ob_dec_attn_kernel<false, 512, true, true>), dim3(agrid), dim3(512), attn_lds, s, at, apf);
            }
            else if (at.st_q && attn_threads == 256 && blind64) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 256, true, false, false, 64>), dim3(agrid), dim3(256), attn_lds, s, at, apf);
            else if (at.st_q && attn_threads == 256) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 256>), dim3(agrid), dim3(256), attn_lds, s, at, apf);
            else if (at.st_q) hipLaunchKernelGGL((ob_dec_attn_kernel<true, 512>), dim3(agrid), dim3(512), attn_lds, s, at, apf);
            else hipLaunchKernelGGL((ob_dec_attn_kernel<false, 512>), dim3(agrid), dim3(512), attn_lds, s, at, apf);
            if ((rc = ob_launch_status("decode_step(attn)"))) return rc;
        }
        // K3: o_proj
        if ((rc = ob_launch_dec_gemv(o, s))) return rc;
        // K4: residual + LN(u_o) -> RMSNorm -> gate, up
        ObGemvArgs gu = {};
        gu.nproj = 2; gu.K = H; gu.prologue = OB_P_RES_LN_RMS;
        if ((rc = ob_fill_proj(gu.p[0], L.gate, st->u_gate, "gate_proj", ts_gate))) return rc;
        if ((rc = ob_fill_proj(gu.p[1], L.up, st->u_up, "up_proj", ts_up))) return rc;
        gu.hres_in = hB; gu.u_prev = (const _Float16 *)st->u_o; gu.hres_out = hA;
        gu.st_prev = ts_o;
        gu.bias_prev = (const _Float16 *)L.o_bias;          // o_proj's bias joins LayerNorm(u_o) in this launch's prologue
        gu.rms_w = (const _Float16 *)L.post_attention_layernorm_w;
        gu.rms_eps = m->rms_eps; gu.ln_eps = m->ln_eps;
        if ((rc = ob_launch_dec_gemv(gu, s))) return rc;
        // K5: silu(LN(gate)) * LN(up) -> down
        ObGemvArgs dn = {};
        dn.nproj = 1; dn.K = I; dn.prologue = OB_P_SWIGLU;
        if ((rc = ob_fill_proj(dn.p[0], L.down, st->u_down, "down_proj", ts_down))) return rc;
        dn.u_gate = (const _Float16 *)st->u_gate; dn.u_up = (const _Float16 *)st->u_up;
        dn.st_gate = ts_gate; dn.st_up = ts_up;
        dn.rms_eps = m->rms_eps; dn.ln_eps = m->ln_eps;
        if ((rc = ob_launch_dec_gemv(dn, s))) return rc;
    }
    // final norm + lm_head + argmax
    ObHeadArgs hd = {};
    hd.hres_in = hA; hd.u_prev = (const _Float16 *)st->u_down; hd.rms_w = (const _Float16 *)m->final_norm_w;
    hd.st_prev = ts_down;
    hd.lm_w = (const _Float16 *)m->lm_head; hd.logits = (_Float16 *)st->logits;
    hd.part_val = st->part_val; hd.part_idx = st->part_idx; hd.hres_out = hB;
    hd.K = H; hd.V = m->vocab; hd.rms_eps = m->rms_eps; hd.ln_eps = m->ln_eps;
    int G = ob_cu_count();
    if (G > 1024) G = 1024;
    const size_t head_lds = (size_t)H * 2 + 64 * 4 + 64 * 4;
    hipLaunchKernelGGL(ob_dec_lmhead_kernel, dim3(G), dim3(OB_DEC_THREADS), head_lds, s, hd);
    if ((rc = ob_launch_status("decode_step(lm_head)"))) return rc;
    hipLaunchKernelGGL(ob_dec_argmax_kernel, dim3(1), dim3(256), 0, s, (const float *)st->part_val,
                       (const int *)st->part_idx, G, st->token, st->pos, st->out_tokens, st->max_out, m->vocab);
    return ob_launch_status("decode_step(argmax)");
}

// ------------------------------------------------------------ K-sharded decode step (config 4) --
// SURVEY.md section 8(e); reference semantics bitnet.py:112-122 with the sum over K split across ranks before :115's rounding.
static int ob_kshard_proj(ObProj &d, const onebit_proj_t &sp, float *z, int64_t n_expect, int64_t k_full, const char *name)
{
    if (!sp.weight || !sp.input_factor || !sp.weight_scale || !z)
        return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null pointer in projection %s", name);
    if (sp.N != n_expect || sp.K <= 0 || sp.K > k_full || sp.K % 128 != 0 || sp.ldw_bytes % 16 != 0 || sp.ldw_bytes < sp.K / 8 ||
        !ob_aligned(sp.weight, 16) || !ob_aligned(sp.input_factor, 16) || !ob_aligned(sp.weight_scale, 16))
        return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: projection %s: the K slice must be a multiple of 128 columns with 16-byte "
                                       "aligned weight rows, input_factor and weight_scale (N = %lld, K = %lld of %lld, pitch %lld)", name,
                       (long long)sp.N, (long long)sp.K, (long long)k_full, (long long)sp.ldw_bytes);
    d.w = (const uint32_t *)sp.weight; d.h = (const _Float16 *)sp.input_factor; d.g = (const _Float16 *)sp.weight_scale;
    d.u = (_Float16 *)z;                       // ZOUT instances: fp32 partial sums
    d.st = nullptr;
    d.N = (int)sp.N; d.K = (int)sp.K; d.ldw = (int)(sp.ldw_bytes / 4);
    return 0;
}

extern "C" int onebit_decode_step_ksharded(const onebit_model_t *m, const onebit_kshard_state_t *st, int32_t l, int32_t segment, void *stream)
{
    if (!m || !st || !m->layers) return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null model/state");
    if (st->struct_size != sizeof(onebit_kshard_state_t))
        return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: state struct_size %llu != %zu (caller built against another ABI: this library is ABI %d)",
                       (unsigned long long)st->struct_size, sizeof(onebit_kshard_state_t), ONEBIT_ABI_VERSION);
    if (m->n_layers <= 0 || m->hidden <= 0 || m->n_heads <= 0 || m->n_kv_heads <= 0 || m->head_dim <= 0 ||
        m->n_heads % m->n_kv_heads != 0 || m->head_dim % 8 != 0 || m->head_dim > 128 || m->hidden % 16 != 0 ||
        m->intermediate % 16 != 0 || m->max_len <= 0 || m->vocab <= 0 || m->hidden > OB_DEC_MAXV * OB_DEC_THREADS * 8 ||
        m->intermediate > OB_DEC_MAXV * OB_DEC_THREADS * 8 || (m->n_heads * m->head_dim) % 16 != 0 || (m->n_kv_heads * m->head_dim) % 16 != 0)
        return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: unsupported model dimensions");
    if (segment < ONEBIT_KSEG_QKV || segment > ONEBIT_KSEG_HEAD) return ob_fail(ONEBIT_E_FLAG, "decode_step_ksharded: unknown segment %d", segment);
    if (segment != ONEBIT_KSEG_HEAD && (l < 0 || l >= m->n_layers)) return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: layer %d", l);
    if (!st->token || !st->pos || !st->hres0 || !st->hres1 || !st->x || !st->u_q || !st->u_k || !st->u_v || !st->attn_out || !st->u_gate ||
        !st->u_up || !st->act || !st->u_down || !st->z_qkv || !st->z_o || !st->z_gu || !st->z_down || !st->logits || !st->part_val ||
        !st->part_idx || !st->tile_stats || !m->embed || !m->final_norm_w || !m->lm_head || !m->rope_cos || !m->rope_sin)
        return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null buffer");
    if (!ob_aligned(st->tile_stats, 16) || !ob_aligned(st->z_qkv, 16) || !ob_aligned(st->z_o, 16) || !ob_aligned(st->z_gu, 16) ||
        !ob_aligned(st->z_down, 16) || !ob_aligned(st->x, 16) || !ob_aligned(st->attn_out, 16) || !ob_aligned(st->act, 16))
        return ob_fail(ONEBIT_E_ALIGN, "decode_step_ksharded: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int H = m->hidden, I = m->intermediate, D = m->head_dim, NQ = m->n_heads * D, NK = m->n_kv_heads * D;
    if (st->k0_hidden < 0 || st->k0_attn < 0 || st->k0_inter < 0 || st->k0_hidden % 128 || st->k0_attn % 128 || st->k0_inter % 128)
        return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: slice origins must be multiples of 128");
    if (st->attn_chunk != 0 && (st->attn_chunk < 0 || st->attn_chunk % 64 != 0 || st->attn_splits < 1 || st->attn_splits > 64 ||
                                (int64_t)st->attn_splits * st->attn_chunk < m->max_len || (st->attn_splits > 1 && !st->attn_scratch) ||
                                (D & (D - 1)) != 0 || D < 16))
        return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: attn_chunk %d needs a multiple of 64, 1..64 attn_splits covering max_len, attn_scratch and a power-of-two head_dim", st->attn_chunk);
    _Float16 *hA = (_Float16 *)st->hres0, *hB = (_Float16 *)st->hres1;
    ObStatsLayout sl = ob_stats_layout(m);
    float *ts = st->tile_stats;
    float *ts_q = ts + sl.off[0], *ts_k = ts + sl.off[1], *ts_v = ts + sl.off[2], *ts_down = ts + sl.off[6];
    int rc;
    auto zg_blocks = [](int n) { return (n + OB_DEC_THREADS * 8 - 1) / (OB_DEC_THREADS * 8); };
    auto launch_zg = [&](ObBZgArgs za, int nseg) -> int {
        int tot = 0;
        for (int i = 0; i < 3; ++i) {
            if (i < nseg) tot += zg_blocks(za.s[i].n);
            else za.s[i] = za.s[nseg - 1];
            za.s[i].blk_end = tot;
        }
        hipLaunchKernelGGL(ob_b_zg_kernel, dim3(tot), dim3(OB_DEC_THREADS), 0, s, za);
        return ob_launch_status("decode_step_ksharded(scale)");
    };
    if (segment == ONEBIT_KSEG_HEAD) {
        const onebit_layer_t &LL = m->layers[m->n_layers - 1];
        ObBZgArgs za = {};
        za.s[0] = {st->z_down, (const _Float16 *)LL.down.weight_scale, (_Float16 *)st->u_down, ts_down, H, 0};
        if (!LL.down.weight_scale) return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null weight_scale");
        if ((rc = launch_zg(za, 1))) return rc;
        ObHeadArgs hd = {};
        hd.hres_in = hA; hd.u_prev = (const _Float16 *)st->u_down; hd.rms_w = (const _Float16 *)m->final_norm_w;
        hd.st_prev = ts_down;
        hd.lm_w = (const _Float16 *)m->lm_head; hd.logits = (_Float16 *)st->logits;
        hd.part_val = st->part_val; hd.part_idx = st->part_idx; hd.hres_out = hB;
        hd.K = H; hd.V = m->vocab; hd.rms_eps = m->rms_eps; hd.ln_eps = m->ln_eps;
        int G = ob_cu_count();
        if (G > 1024) G = 1024;
        const size_t head_lds = (size_t)H * 2 + 64 * 4 + 64 * 4;
        hipLaunchKernelGGL(ob_dec_lmhead_kernel, dim3(G), dim3(OB_DEC_THREADS), head_lds, s, hd);
        if ((rc = ob_launch_status("decode_step_ksharded(lm_head)"))) return rc;
        hipLaunchKernelGGL(ob_dec_argmax_kernel, dim3(1), dim3(256), 0, s, (const float *)st->part_val,
                           (const int *)st->part_idx, G, st->token, st->pos, st->out_tokens, st->max_out, m->vocab);
        return ob_launch_status("decode_step_ksharded(argmax)");
    }
    const onebit_layer_t &L = m->layers[l];
    if (!L.k_cache || !L.v_cache || !L.input_layernorm_w || !L.post_attention_layernorm_w)
        return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null buffer in layer %d", l);
    const bool qkv_bias = L.q_bias || L.k_bias || L.v_bias;
    if (qkv_bias && !(L.q_bias && L.k_bias && L.v_bias))
        return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: layer %d has some but not all of q_bias / k_bias / v_bias", l);
    { const int rcb = ob_check_bias_align(L, "decode_step_ksharded", l); if (rcb) return rcb; }
    ObGemvArgs a = {};
    a.prologue = OB_P_PLAIN; a.zout = 1; a.rms_eps = m->rms_eps; a.ln_eps = m->ln_eps;
    ObBNormArgs na = {};
    na.H = H; na.rms_eps = m->rms_eps; na.ln_eps = m->ln_eps; na.x = (_Float16 *)st->x;
    switch (segment) {
    case ONEBIT_KSEG_QKV: {
        // residual (+ LayerNorm of the previous layer's REDUCED down_proj sums) + input RMSNorm, once; then q | k | v on the slice
        na.embed = (const _Float16 *)m->embed; na.tokens = st->token; na.hres_in = hA; na.hres_out = hB;
        na.rms_w = (const _Float16 *)L.input_layernorm_w;
        if (l > 0) { na.z0 = st->z_down; na.g_prev = (const _Float16 *)m->layers[l - 1].down.weight_scale;
                     if (!na.g_prev) return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null weight_scale"); }
        if (l == 0) OB_LAUNCH_NORM(true, H, dim3(1), s, na);
        else OB_LAUNCH_NORM(false, H, dim3(1), s, na);
        if ((rc = ob_launch_status("decode_step_ksharded(norm)"))) return rc;
        a.nproj = 3; a.K = (int)L.q.K;
        if (L.k.K != L.q.K || L.v.K != L.q.K || st->k0_hidden + L.q.K > H)
            return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: q / k / v slices of layer %d differ or leave the input vector", l);
        if ((rc = ob_kshard_proj(a.p[0], L.q, st->z_qkv, NQ, H, "q_proj"))) return rc;
        if ((rc = ob_kshard_proj(a.p[1], L.k, st->z_qkv + NQ, NK, H, "k_proj"))) return rc;
        if ((rc = ob_kshard_proj(a.p[2], L.v, st->z_qkv + NQ + NK, NK, H, "v_proj"))) return rc;
        a.xin = (const _Float16 *)st->x + st->k0_hidden;
        return ob_launch_dec_gemv(a, s);
    }
    case ONEBIT_KSEG_ATTN_O: {
        // the attention workgroups form u = fp16(fp16(z) * g) of the reduced q | k | v sums themselves and recompute the LayerNorm
        // statistics from them (ZIN instances: 92 KB of L2 reads per head workgroup instead of one more launch per layer)
        if (!L.q.weight_scale || !L.k.weight_scale || !L.v.weight_scale) return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null weight_scale");
        ObAttnArgs at = {};
        at.u_q = (const _Float16 *)st->u_q; at.u_k = (const _Float16 *)st->u_k; at.u_v = (const _Float16 *)st->u_v;
        at.cos = (const _Float16 *)m->rope_cos; at.sin = (const _Float16 *)m->rope_sin;
        at.kcache = (_Float16 *)L.k_cache; at.vcache = (_Float16 *)L.v_cache; at.out = (_Float16 *)st->attn_out;
        at.pos = st->pos; at.H = m->n_heads; at.Hkv = m->n_kv_heads; at.D = D; at.max_len = m->max_len;
        at.ln_eps = m->ln_eps;
        at.z_q = st->z_qkv; at.z_k = st->z_qkv + NQ; at.z_v = st->z_qkv + NQ + NK;
        at.g_q = (const _Float16 *)L.q.weight_scale; at.g_k = (const _Float16 *)L.k.weight_scale; at.g_v = (const _Float16 *)L.v.weight_scale;
        at.b_q = (const _Float16 *)L.q_bias; at.b_k = (const _Float16 *)L.k_bias; at.b_v = (const _Float16 *)L.v_bias;
        if (st->attn_chunk > 0) {
            // key-block form: u = fp16(fp16(z) * g) of the three reduced sums + their LayerNorm partials in one row launch, then the
            // attention launch of the other engines' long-context route (split over the cached tokens, q / k / v formed inside)
            ObBZgArgs za = {};
            za.s[0] = {st->z_qkv, (const _Float16 *)L.q.weight_scale, (_Float16 *)st->u_q, ts_q, NQ, 0};
            za.s[1] = {st->z_qkv + NQ, (const _Float16 *)L.k.weight_scale, (_Float16 *)st->u_k, ts_k, NK, 0};
            za.s[2] = {st->z_qkv + NQ + NK, (const _Float16 *)L.v.weight_scale, (_Float16 *)st->u_v, ts_v, NK, 0};
            if ((rc = launch_zg(za, 3))) return rc;
            const size_t sbytes = onebit_attention_decode_scratch_bytes(1, m->n_heads, st->attn_splits);
            if ((rc = onebit_attention_decode_rows_fused(st->u_q, st->u_k, st->u_v, ts_q, ts_k, ts_v, L.q_bias, L.k_bias, L.v_bias, m->rope_cos,
                                                         m->rope_sin, L.k_cache, L.v_cache, st->attn_out, nullptr, nullptr, st->pos, 1, m->n_heads,
                                                         m->n_kv_heads, D, 1, m->max_len, m->max_len, st->attn_chunk, st->attn_splits, m->ln_eps,
                                                         st->attn_scratch, sbytes, s)))
                return rc;
        } else {
            const size_t attn_lds = 512 + 3 * 128 * 2 + (size_t)OB_ATTN_WAVES * 128 * 4 + (size_t)4 * m->max_len;
            if (attn_lds > 64 * 1024)
                return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: max_len %d too large for the one-workgroup attention kernel (set attn_chunk)", m->max_len);
            if (qkv_bias) hipLaunchKernelGGL((ob_dec_attn_kernel<false, 512, true, true, true>), dim3(m->n_heads), dim3(512), attn_lds, s, at, ObPfPlan{});
            else hipLaunchKernelGGL((ob_dec_attn_kernel<false, 512, true, false, true>), dim3(m->n_heads), dim3(512), attn_lds, s, at, ObPfPlan{});
            if ((rc = ob_launch_status("decode_step_ksharded(attn)"))) return rc;
        }
        a.nproj = 1; a.K = (int)L.o.K;
        if (st->k0_attn + L.o.K > NQ) return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: o_proj slice of layer %d leaves the input vector", l);
        if ((rc = ob_kshard_proj(a.p[0], L.o, st->z_o, H, NQ, "o_proj"))) return rc;
        a.xin = (const _Float16 *)st->attn_out + st->k0_attn;
        return ob_launch_dec_gemv(a, s);
    }
    case ONEBIT_KSEG_GATE_UP: {
        na.hres_in = hB; na.hres_out = hA; na.z0 = st->z_o; na.g_prev = (const _Float16 *)L.o.weight_scale;
        na.bias_prev = (const _Float16 *)L.o_bias;
        na.rms_w = (const _Float16 *)L.post_attention_layernorm_w;
        if (!na.g_prev) return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null weight_scale");
        OB_LAUNCH_NORM(false, H, dim3(1), s, na);
        if ((rc = ob_launch_status("decode_step_ksharded(norm2)"))) return rc;
        a.nproj = 2; a.K = (int)L.gate.K;
        if (L.up.K != L.gate.K || L.gate.K != L.q.K || st->k0_hidden + L.gate.K > H)
            return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: gate / up slices of layer %d differ from q's or leave the input vector", l);
        if ((rc = ob_kshard_proj(a.p[0], L.gate, st->z_gu, I, H, "gate_proj"))) return rc;
        if ((rc = ob_kshard_proj(a.p[1], L.up, st->z_gu + I, I, H, "up_proj"))) return rc;
        a.xin = (const _Float16 *)st->x + st->k0_hidden;
        return ob_launch_dec_gemv(a, s);
    }
    default: {   // ONEBIT_KSEG_DOWN
        if (!L.gate.weight_scale || !L.up.weight_scale) return ob_fail(ONEBIT_E_ARG, "decode_step_ksharded: null weight_scale");
        ObBSwigluArgs sa = {};                 // the SwiGLU row kernel rounds and scales the reduced gate | up sums itself
        sa.act = (_Float16 *)st->act; sa.I = I; sa.ln_eps = m->ln_eps;
        sa.z_gate = st->z_gu; sa.z_up = st->z_gu + I;
        sa.g_gate = (const _Float16 *)L.gate.weight_scale; sa.g_up = (const _Float16 *)L.up.weight_scale;
        OB_LAUNCH_SWIGLU(I, dim3(1), s, sa);
        if ((rc = ob_launch_status("decode_step_ksharded(swiglu)"))) return rc;
        a.nproj = 1; a.K = (int)L.down.K;
        if (st->k0_inter + L.down.K > I) return ob_fail(ONEBIT_E_SHAPE, "decode_step_ksharded: down_proj slice of layer %d leaves the input vector", l);
        if ((rc = ob_kshard_proj(a.p[0], L.down, st->z_down, H, I, "down_proj"))) return rc;
        a.xin = (const _Float16 *)st->act + st->k0_inter;
        return ob_launch_dec_gemv(a, s);
    }
    }
}

// ------------------------------------------------------------ train-mode layer --
// onebit_train_forward / onebit_train_backward: bitnet.py:14-28 (SignSTE) and :58-68 (BitLinear.forward)

template <typename TI, bool RCA, bool RCB, int TXA, int TXB, int EPI>
static void ob_launch_tgemm(const ObTGemmArgs &a, hipStream_t s)
{
    if constexpr (std::is_same<TI, _Float16>::value) {      // gfx950 tiling for fp16 (ob_train.h); fp32 keeps the 64 x 64 form
        static const bool legacy = getenv("OB_TRAIN_GEMM64") != nullptr;
        if (!legacy) {
            const dim3 grid((unsigned)((a.N + OB_TG2_B - 1) / OB_TG2_B), (unsigned)((a.M + OB_TG2_B - 1) / OB_TG2_B));
            static bool attr_set[OB_MAX_DEVICES] = {};
            ob_set_max_lds_once(ob_tgemm128_f16_kernel<RCA, RCB, TXA, TXB, EPI>, attr_set, OB_TG2_LDS);
            hipLaunchKernelGGL((ob_tgemm128_f16_kernel<RCA, RCB, TXA, TXB, EPI>), grid, dim3(OB_TG_THREADS), OB_TG2_LDS, s, a);
            return;
        }
    }
    const dim3 grid((unsigned)((a.N + OB_TG_BN - 1) / OB_TG_BN), (unsigned)((a.M + OB_TG_BM - 1) / OB_TG_BM));
    hipLaunchKernelGGL((ob_tgemm_kernel<TI, RCA, RCB, TXA, TXB, EPI>), grid, dim3(OB_TG_THREADS), 0, s, a);
}

static size_t ob_train_align(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t onebit_train_workspace_bytes(int64_t T, int64_t K, int64_t N, int dtype)
{
    if (T <= 0 || K <= 0 || N <= 0) return 0;
    const size_t sz = dtype == ONEBIT_F32 ? 4 : 2;
    return ob_train_align((size_t)T * N * sz) + ob_train_align((size_t)T * K * sz) + ob_train_align((size_t)T * 2 * sizeof(float)) +
           ob_train_align((size_t)OB_TC_SLICES * (2 * (size_t)N > (size_t)K ? 2 * (size_t)N : (size_t)K) * sizeof(float));     // column-sum partials
}

static int ob_train_check(const char *what, int64_t T, int64_t K, int64_t N, int dtype)
{
    if (T < 0 || K < 0 || N < 0) return ob_fail(ONEBIT_E_ARG, "%s: negative size", what);
    if (dtype != ONEBIT_F16 && dtype != ONEBIT_F32) return ob_fail(ONEBIT_E_DTYPE, "%s: dtype %d", what, dtype);
    if (T > 0x7fffffffLL || K > 0x7fffffffLL || N > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "%s: dimension too large", what);
    return 0;
}

template <typename TI>
static int ob_train_forward_t(const void *x, const void *w, const void *h, const void *g, const void *bias, void *y, void *z,
                              float *stats, int64_t T, int64_t K, int64_t N, float eps, hipStream_t s)
{
    ObTGemmArgs a = {};
    a.A = x; a.lda = K; a.va = h;            // A(t, k) = x[t, k] * h[k]
    a.B = w; a.ldb = K;                      // B(n, k) = sign(W[n, k])
    a.C = z; a.ldc = N; a.M = (int)T; a.N = (int)N; a.R = (int)K;
    ob_launch_tgemm<TI, true, true, OB_TX_SCALE_R, OB_TX_SIGN, OB_TE_PLAIN>(a, s);
    int rc = ob_launch_status("train_forward(gemm)");
    if (rc) return rc;
    hipLaunchKernelGGL((ob_train_ln_fwd_kernel<TI>), dim3((unsigned)T), dim3(256), 0, s, (const TI *)z, (const TI *)g, (const TI *)bias,
                       (TI *)y, stats, (int)N, eps);
    return ob_launch_status("train_forward(layernorm)");
}

extern "C" int onebit_train_forward(const void *x, const void *w_latent, const void *h, const void *g, const void *bias_or_null,
                                    void *y, void *z_save, float *ln_stats, int64_t T, int64_t K, int64_t N, int dtype,
                                    float ln_eps, void *stream)
{
    int rc = ob_train_check("train_forward", T, K, N, dtype);
    if (rc) return rc;
    if (T == 0 || N == 0) return 0;
    if (K == 0) return ob_fail(ONEBIT_E_SHAPE, "train_forward: in_features = 0");
    if (!x || !w_latent || !h || !g || !y || !z_save || !ln_stats) return ob_fail(ONEBIT_E_ARG, "train_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    return dtype == ONEBIT_F16 ? ob_train_forward_t<_Float16>(x, w_latent, h, g, bias_or_null, y, z_save, ln_stats, T, K, N, ln_eps, s)
                               : ob_train_forward_t<float>(x, w_latent, h, g, bias_or_null, y, z_save, ln_stats, T, K, N, ln_eps, s);
}

template <typename TI>
static int ob_train_backward_t(const void *gy, const void *x, const void *w, const void *h, const void *g, const void *z,
                               const float *stats, void *gx, void *gw, void *gh, void *gg, void *gbias, char *ws,
                               int64_t T, int64_t K, int64_t N, hipStream_t s)
{
    TI *gz = (TI *)ws;
    TI *ga = (TI *)(ws + ob_train_align((size_t)T * N * sizeof(TI)));
    float *rowc = (float *)(ws + ob_train_align((size_t)T * N * sizeof(TI)) + ob_train_align((size_t)T * K * sizeof(TI)));
    float *part = (float *)((char *)rowc + ob_train_align((size_t)T * 2 * sizeof(float)));     // [OB_TC_SLICES][2 N | K] column-sum partials
    int rc;
    // 1. through the LayerNorm and * g: gz [T, N]; column sums gg, gbias
    hipLaunchKernelGGL((ob_train_ln_bwd_kernel<TI>), dim3((unsigned)T), dim3(256), 0, s, (const TI *)gy, (const TI *)z, (const TI *)g, stats, gz, rowc, (int)N);
    if ((rc = ob_launch_status("train_backward(layernorm)"))) return rc;
    hipLaunchKernelGGL((ob_train_cols_ln_kernel<TI>), dim3((unsigned)((N + 63) / 64), OB_TC_SLICES), dim3(256), 0, s, (const TI *)gy, (const TI *)z, (const TI *)g, stats,
                       (const float *)rowc, part, (int)T, (int)N);
    hipLaunchKernelGGL((ob_train_cols_finish_kernel<TI, 2>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const float *)part, (TI *)gg, (TI *)gbias, (int)N);
    if ((rc = ob_launch_status("train_backward(gg)"))) return rc;
    // 2. ga [T, K] = gz . sign(W);  gx = ga * h;  gh = sum_t ga * x
    {
        ObTGemmArgs a = {};
        a.A = gz; a.lda = N;                 // A(t, n) = gz[t, n]            (n contiguous)
        a.B = w; a.ldb = K;                  // B(k, n) = sign(W[n, k])        (k = output index, contiguous)
        a.C = ga; a.C2 = gx; a.vc = h; a.ldc = K; a.M = (int)T; a.N = (int)K; a.R = (int)N;
        ob_launch_tgemm<TI, true, false, OB_TX_NONE, OB_TX_SIGN, OB_TE_GX>(a, s);
        if ((rc = ob_launch_status("train_backward(grad input)"))) return rc;
        hipLaunchKernelGGL((ob_train_cols_gh_kernel<TI>), dim3((unsigned)((K + 63) / 64), OB_TC_SLICES), dim3(256), 0, s, (const TI *)ga, (const TI *)x, part, (int)T, (int)K);
        hipLaunchKernelGGL((ob_train_cols_finish_kernel<TI, 1>), dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, (const float *)part, (TI *)gh, (TI *)nullptr, (int)K);
        if ((rc = ob_launch_status("train_backward(gh)"))) return rc;
    }
    // 3. gW [N, K] = (gz^T . (x * h)) * (1.001 - tanh(W)^2)
    {
        ObTGemmArgs a = {};
        a.A = gz; a.lda = N;                 // A(n, t) = gz[t, n]            (n = output index, contiguous)
        a.B = x; a.ldb = K; a.vb = h;        // B(k, t) = x[t, k] * h[k]      (k = output index, contiguous)
        a.C = gw; a.vc = w; a.ldc = K; a.M = (int)N; a.N = (int)K; a.R = (int)T;
        ob_launch_tgemm<TI, false, false, OB_TX_NONE, OB_TX_SCALE_I, OB_TE_STE>(a, s);
        if ((rc = ob_launch_status("train_backward(grad weight)"))) return rc;
    }
    return 0;
}

extern "C" int onebit_train_backward(const void *gy, const void *x, const void *w_latent, const void *h, const void *g,
                                     const void *z_save, const float *ln_stats, void *gx, void *gw, void *gh, void *gg,
                                     void *gbias_or_null, void *workspace, size_t workspace_bytes, int64_t T, int64_t K,
                                     int64_t N, int dtype, void *stream)
{
    int rc = ob_train_check("train_backward", T, K, N, dtype);
    if (rc) return rc;
    if (T == 0 || N == 0 || K == 0) return ob_fail(ONEBIT_E_SHAPE, "train_backward: empty problem");
    if (!gy || !x || !w_latent || !h || !g || !z_save || !ln_stats || !gx || !gw || !gh || !gg)
        return ob_fail(ONEBIT_E_ARG, "train_backward: null pointer");
    if (!workspace || workspace_bytes < onebit_train_workspace_bytes(T, K, N, dtype) || !ob_aligned(workspace, 16))
        return ob_fail(ONEBIT_E_WSPACE, "train_backward: needs %zu bytes of 16-byte aligned workspace", onebit_train_workspace_bytes(T, K, N, dtype));
    hipStream_t s = (hipStream_t)stream;
    return dtype == ONEBIT_F16 ? ob_train_backward_t<_Float16>(gy, x, w_latent, h, g, z_save, ln_stats, gx, gw, gh, gg, gbias_or_null, (char *)workspace, T, K, N, s)
                               : ob_train_backward_t<float>(gy, x, w_latent, h, g, z_save, ln_stats, gx, gw, gh, gg, gbias_or_null, (char *)workspace, T, K, N, s);
}

// ------------------------------------------------------------ prefill attention --
extern "C" int onebit_attention_prefill(const void *q, const void *k_cache, const void *v_cache, void *o, const void *h_next,
                                        int64_t B, int64_t S, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim,
                                        int64_t past_len, int64_t max_len, void *stream)
{
    if (B < 0 || S < 0 || n_heads <= 0 || n_kv_heads <= 0 || past_len < 0) return ob_fail(ONEBIT_E_ARG, "attention_prefill: bad size");
    if ((head_dim != 64 && head_dim != 128) || n_heads % n_kv_heads != 0)
        return ob_fail(ONEBIT_E_SHAPE, "attention_prefill: head_dim %d (64 or 128), heads %d / %d", head_dim, n_heads, n_kv_heads);
    if (past_len + S > max_len) return ob_fail(ONEBIT_E_SHAPE, "attention_prefill: %lld + %lld tokens beyond the cache (%lld)",
                                               (long long)past_len, (long long)S, (long long)max_len);
    if (B == 0 || S == 0) return 0;
    if (!q || !k_cache || !v_cache || !o) return ob_fail(ONEBIT_E_ARG, "attention_prefill: null pointer");
    if (!ob_aligned(q, 16) || !ob_aligned(k_cache, 16) || !ob_aligned(v_cache, 16) || !ob_aligned(o, 8) || (h_next && !ob_aligned(h_next, 8)))
        return ob_fail(ONEBIT_E_ALIGN, "attention_prefill: q / k / v must be 16-byte aligned");
    const int64_t nmb = (S + OB_FL_BM - 1) / OB_FL_BM;
    if (nmb * n_heads * B > 0x7fffffffLL || max_len > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "attention_prefill: dimension too large");
    ObFlashArgs a = {(const _Float16 *)q, (const _Float16 *)k_cache, (const _Float16 *)v_cache, (_Float16 *)o, (const _Float16 *)h_next,
                     (int)S, n_heads, n_kv_heads, (int)max_len, (int)past_len, 1.4426950408889634f / sqrtf((float)head_dim), (int)nmb};
    const dim3 grid((unsigned)(((nmb + 1) / 2) * n_heads * B));      // a workgroup takes a heavy and a light query block
    if (head_dim == 128) hipLaunchKernelGGL((ob_flash_fwd_kernel<128>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((ob_flash_fwd_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return ob_launch_status("attention_prefill");
}
