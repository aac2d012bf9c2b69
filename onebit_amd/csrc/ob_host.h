// Host-side helpers shared by the translation units of libonebit_hip.so (onebit_hip.hip, onebit_mixed.hip).
// Nothing here is part of the C ABI: the cross-unit functions have hidden visibility.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/onebit.h"

#define OB_HIDDEN __attribute__((visibility("hidden")))

// thread-local last-error message + return code (defined in onebit_hip.hip)
OB_HIDDEN int ob_fail(int code, const char *fmt, ...);
OB_HIDDEN int ob_launch_status(const char *what);
OB_HIDDEN int ob_cu_count();

static inline bool ob_aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

// Function attributes and device properties are per DEVICE, not per process: one process may
// drive several GPUs (hipSetDevice between calls), so "already done" is keyed by the current device.
#define OB_MAX_DEVICES 64
static inline int ob_device_index()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OB_MAX_DEVICES) dev = 0;
    return dev;
}
template <typename F>
static inline void ob_set_max_lds_once(F kernel, bool (&done)[OB_MAX_DEVICES], int bytes)
{
    const int dev = ob_device_index();
    if (!done[dev]) {
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done[dev] = true;
    }
}

// ---- launchers of onebit_hip.hip that the mixed prefill + decode step (onebit_mixed.hip) drives directly ----------------
// Row kernel of the batched / prefill glue (ob_b_norm_kernel) in every form the mixed step needs:
//   embed != NULL:  r = embed[tokens[t]]                        (layer 0: modeling_bitllama.py:1275)
//   else:           r = hres_in[t'] + LayerNorm(u_prev[t']) (+ bias_prev),  t' = rows ? rows[t] : t   (gathered final norm)
//   hres_out[t] = r;  x[t] = RMSNorm(r) * rms_w;  x_scaled[i][t] = fp16(x * h_next[i])
struct ObRowsNormCall {
    const void *embed; const int32_t *tokens;
    const void *hres_in, *u_prev, *bias_prev, *rms_w;
    void *hres_out, *x;
    const void *h_next[3]; void *x_scaled[3]; int n_scaled;
    const int32_t *rows;
    int64_t T; int H; float rms_eps, ln_eps;
    // instead of u_prev: the previous projection as up to four K-slices' fp32 sums (ob_gemm3_ksplit) + its weight_scale:
    // u = fp16(fp16(z0 + z1 (+ z2 (+ z3))) * g_prev) (bitnet.py:115-116) is formed by the row kernel
    const float *z0, *z1; const void *g_prev;
    const float *z2, *z3;
};
OB_HIDDEN int ob_rows_norm(const ObRowsNormCall &c, hipStream_t s);
// fp16 lm_head for B <= 64 rows + greedy token per row (ob_b_lmhead_kernel + ob_b_argmax_kernel)
OB_HIDDEN int ob_lm_head_argmax(const void *x, const void *lm_head, void *logits_or_null, float *part_val, int32_t *part_idx,
                                int32_t *next_tokens, int B, int H, int V, hipStream_t s);
// up to three projections that share T and K in ONE LDS-DMA skinny launch (2 <= T <= 64) on their pre-scaled rows a[i] [T, K]:
// u[i] = fp16(fp16(W_i . a_i) * g_i).  Returns ONEBIT_E_SHAPE when a projection is not eligible (the caller falls back to
// onebit_linear_forward per projection).
OB_HIDDEN int ob_sk3_multi(const onebit_proj_t *const *ps, void *const *us, const void *const *as, int np, int64_t T, hipStream_t s);
// the projection's part of that eligibility (shape, pitch, the weights' alignment; rows assumed 16-byte aligned, 2 <= T <= 64)
OB_HIDDEN bool ob_sk3_proj_ok(const onebit_proj_t &p);
// up to three projections sharing T, K and the packed row pitch in ONE launch of the LDS-DMA prefill GEMM (ob_gemm3g_f16_kernel) on
// their pre-scaled rows a[i] [T, K]: u[i] = fp16(fp16(W_i . a_i) * g_i).  ONEBIT_E_SHAPE when the group is not eligible (the caller
// falls back to onebit_linear_forward per projection).
OB_HIDDEN int ob_gemm3_grouped(const onebit_proj_t *const *ps, void *const *us, const void *const *as, int np, int64_t T, hipStream_t s);
// the eligibility test of ob_gemm3_grouped alone (shapes, pitches, the weights' alignment, the GROUP's tile count against the CU count;
// rows / outputs assumed 16-byte aligned): a caller that is about to choose between pre-scaled rows for the group and plain rows asks first
OB_HIDDEN bool ob_gemm3_group_ok(const onebit_proj_t *const *ps, int np, int64_t T);
// 256 x 128 tiles of that launch (sum over the members) and the workgroup slots of one round (two workgroups per CU)
OB_HIDDEN int64_t ob_gemm3_group_tiles(const onebit_proj_t *const *ps, int np, int64_t T);
OB_HIDDEN int ob_gemm3_slots();
// ONE projection on its pre-scaled rows a [T, K] as two to four K-slices in one launch of the LDS-DMA GEMM (fp32 sums z[i] [T, N], added and
// scaled by the consuming row kernel): a hidden-width projection at a few hundred rows has too few 256 x 128 tiles for the chip (13B, 543 rows:
// 100), its slices have that many times more.  ob_gemm3_ksplit_n: 0 when the projection alone is eligible for the LDS-DMA GEMM or no
// slicing fills two thirds of the CUs, else the number of slices (the fewest that do).
OB_HIDDEN int ob_gemm3_ksplit_n(const onebit_proj_t &p, int64_t T);
OB_HIDDEN int ob_gemm3_ksplit(const onebit_proj_t &p, const void *a, float *const *z, int ns, int64_t T, hipStream_t s);
