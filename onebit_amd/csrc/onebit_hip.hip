// libonebit_hip.so -- C ABI (include/onebit.h) over the gfx950 kernels.  No torch types, no
// allocation, no synchronisation: every call validates its arguments and enqueues kernels on
// the caller's stream.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/onebit.h"
#include "ob_linear.h"
#include "ob_pack.h"

static thread_local char g_err[256] = "";

static int ob_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int ob_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ob_fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

static inline bool ob_aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

extern "C" int onebit_abi_version(void) { return ONEBIT_ABI_VERSION; }
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

// z (fp32 partial, PARTIAL) or u (fp16) for T tokens.
template <bool PARTIAL>
static void ob_launch_mm16(const void *packed, int64_t ldw_bytes, const void *x, int64_t ldx,
                           const void *h, const void *g, void *u, float *zp, int64_t T, int64_t K,
                           int64_t N, hipStream_t s)
{
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

extern "C" size_t onebit_linear_workspace_bytes(int64_t T, int64_t K, int64_t N, int dtype)
{
    if (T <= 0 || N <= 0) return 0;
    // F32: fp32 z is staged in y itself.  F16 on the MFMA path: u is staged in y.  F16 shapes the
    // MFMA path cannot take (K % 32 != 0) stage fp32 z in the workspace.
    if (dtype == ONEBIT_F16 && K % 32 != 0) return (size_t)T * (size_t)N * sizeof(float);
    return 0;
}

extern "C" int onebit_linear_forward(const void *packed, int64_t ldw_bytes, const void *x,
                                     const void *h, const void *g, const void *bias, void *y,
                                     void *u_or_null, void *workspace, size_t workspace_bytes,
                                     int64_t T, int64_t K, int64_t N, int dtype, float ln_eps,
                                     unsigned flags, void *stream)
{
    int rc = ob_check_linear("linear_forward", packed, ldw_bytes, x, h, T, K, N, dtype);
    if (rc) return rc;
    if (flags & ~ONEBIT_FLAG_SKIP_LN) return ob_fail(ONEBIT_E_FLAG, "linear_forward: unknown flags 0x%x", flags);
    if (workspace_bytes < onebit_linear_workspace_bytes(T, K, N, dtype))
        return ob_fail(ONEBIT_E_WSPACE, "linear_forward: workspace too small");
    if (T == 0 || N == 0) return 0;
    if (!g || !y) return ob_fail(ONEBIT_E_ARG, "linear_forward: null pointer");
    if (!ob_aligned(g, 2) || !ob_aligned(y, 16)) return ob_fail(ONEBIT_E_ALIGN, "linear_forward: y must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int skip = (flags & ONEBIT_FLAG_SKIP_LN) ? 1 : 0;
    if (dtype == ONEBIT_F16 && (K == 0 || ob_mfma_ok(packed, ldw_bytes, K, K, dtype))) {
        _Float16 *ubuf = (_Float16 *)(u_or_null ? u_or_null : y);
        if (K == 0) {
            (void)hipMemsetAsync(ubuf, 0, (size_t)T * N * 2, s);
        } else {
            ob_launch_mm16<false>(packed, ldw_bytes, x, K, h, g, ubuf, nullptr, T, K, N, s);
            rc = ob_launch_status("linear_forward(mm16)");
            if (rc) return rc;
        }
        if (skip && ubuf == y) return 0;
        hipLaunchKernelGGL((ob_layernorm_kernel<_Float16, false>), dim3((unsigned)T), dim3(256), 0, s,
                           (const float *)nullptr, (const _Float16 *)ubuf, (const _Float16 *)g,
                           (const _Float16 *)bias, (_Float16 *)y, (_Float16 *)nullptr, (int)N, ln_eps, skip);
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
        hipLaunchKernelGGL((ob_layernorm_kernel<_Float16, true>), dim3((unsigned)T), dim3(256), 0, s,
                           (const float *)workspace, (const _Float16 *)nullptr, (const _Float16 *)g,
                           (const _Float16 *)bias, (_Float16 *)y, (_Float16 *)u_or_null, (int)N, ln_eps, skip);
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

extern "C" int onebit_matmul_partial(const void *packed, int64_t ldw_bytes, const void *x,
                                     int64_t ldx, const void *h, float *zp, int64_t T, int64_t K,
                                     int64_t N, int dtype, void *stream)
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
    if (ob_mfma_ok(packed, ldw_bytes, ldx, K, dtype))
        ob_launch_mm16<true>(packed, ldw_bytes, x, ldx, h, nullptr, nullptr, zp, T, K, N, s);
    else if (dtype == ONEBIT_F16)
        ob_launch_simple<_Float16>(packed, ldw_bytes, x, ldx, h, zp, T, K, N, s);
    else
        ob_launch_simple<float>(packed, ldw_bytes, x, ldx, h, zp, T, K, N, s);
    return ob_launch_status("matmul_partial");
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
        hipLaunchKernelGGL((ob_layernorm_kernel<_Float16, true>), dim3((unsigned)T), dim3(256), 0, s, z,
                           (const _Float16 *)nullptr, (const _Float16 *)g, (const _Float16 *)bias,
                           (_Float16 *)y, (_Float16 *)u_or_null, (int)N, ln_eps, skip);
    else
        hipLaunchKernelGGL((ob_layernorm_kernel<float, true>), dim3((unsigned)T), dim3(256), 0, s, z,
                           (const float *)nullptr, (const float *)g, (const float *)bias, (float *)y,
                           (float *)u_or_null, (int)N, ln_eps, skip);
    return ob_launch_status("scale_layernorm");
}
