// libonebit_hip.so, second translation unit (round 6): attention over RAGGED token rows and the native mixed
// prefill + decode step of continuous batching (BASELINE config 5).
//   onebit_attention_ragged        causal flash attention (ob_flash.h, RAGGED form) over prompt chunks of several sequences
//   onebit_attention_decode_rows   split-KV decode attention (ob_fdec.h) for single-token rows, any context length
//   onebit_mixed_step              one scheduler step: token rows of ALL scheduled items concatenated -> every 1-bit projection
//                                  ONE GEMM over all rows, fused row glue, ragged attention, lm_head on the rows that sample
// The reference has no counterpart for the batching itself; the arithmetic per row is modeling_bitllama.py:869-918 (layer),
// :487-585 (attention with past), bitnet.py:112-122 (projection), generation/utils.py:2540 (greedy token).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <math.h>
#include <stdlib.h>

#include "../../include/onebit.h"
#include "ob_host.h"
#include "ob_flash.h"
#include "ob_fdec.h"

// ------------------------------------------------------------------------------------------------ grouped projections --
extern "C" int onebit_linear_group_prescaled(const onebit_proj_t *projs, void *const *u, const void *const *a, int32_t n_proj, int64_t T, void *stream)
{
    if (!projs || !u || !a || n_proj < 1 || n_proj > 3 || T < 0) return ob_fail(ONEBIT_E_ARG, "linear_group_prescaled: bad arguments");
    if (T == 0) return 0;
    const onebit_proj_t *ps[3] = {&projs[0], &projs[n_proj > 1 ? 1 : 0], &projs[n_proj > 2 ? 2 : 0]};
    if (T >= 2 && T <= 64) return ob_sk3_multi(ps, u, a, n_proj, T, (hipStream_t)stream);
    if (n_proj < 2) return ob_fail(ONEBIT_E_SHAPE, "linear_group_prescaled: one projection at T > 64: call onebit_linear_forward");
    return ob_gemm3_grouped(ps, u, a, n_proj, T, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ ragged prefill attention --
extern "C" int onebit_attention_ragged(const void *q, const void *k_cache, const void *v_cache, void *o, const void *h_next,
                                       const onebit_seg_t *segs, int32_t n_seg, int32_t n_heads, int32_t n_kv_heads,
                                       int32_t head_dim, int64_t n_slots, int64_t max_len, void *stream)
{
    if (n_seg < 0 || n_heads <= 0 || n_kv_heads <= 0 || n_slots <= 0 || max_len <= 0) return ob_fail(ONEBIT_E_ARG, "attention_ragged: bad size");
    if ((head_dim != 64 && head_dim != 128) || n_heads % n_kv_heads != 0)
        return ob_fail(ONEBIT_E_SHAPE, "attention_ragged: head_dim %d (64 or 128), heads %d / %d", head_dim, n_heads, n_kv_heads);
    if (n_seg == 0) return 0;
    if (!q || !k_cache || !v_cache || !o || !segs) return ob_fail(ONEBIT_E_ARG, "attention_ragged: null pointer");
    if (!ob_aligned(q, 16) || !ob_aligned(k_cache, 16) || !ob_aligned(v_cache, 16) || !ob_aligned(o, 8) || (h_next && !ob_aligned(h_next, 8)))
        return ob_fail(ONEBIT_E_ALIGN, "attention_ragged: q / k / v must be 16-byte aligned");
    if (max_len > 0x7fffffffLL / (2 * head_dim)) return ob_fail(ONEBIT_E_ARG, "attention_ragged: max_len too large");
    for (int i = 0; i < n_seg; ++i) {
        const onebit_seg_t &g = segs[i];
        if (g.n < 1 || g.row0 < 0 || g.slot < 0 || g.slot >= n_slots || g.past < 0 || (int64_t)g.past + g.n > max_len)
            return ob_fail(ONEBIT_E_SHAPE, "attention_ragged: segment %d (rows %d + %d, slot %d, past %d) outside %lld slots x %lld positions",
                           i, g.row0, g.n, g.slot, g.past, (long long)n_slots, (long long)max_len);
    }
    for (int s0 = 0; s0 < n_seg; s0 += OB_FL_MAXSEG) {
        const int ns = std::min(n_seg - s0, OB_FL_MAXSEG);
        ObFlashRaggedArgs a = {};
        a.a.q = (const _Float16 *)q; a.a.k = (const _Float16 *)k_cache; a.a.v = (const _Float16 *)v_cache; a.a.o = (_Float16 *)o;
        a.a.h_next = (const _Float16 *)h_next; a.a.H = n_heads; a.a.Hkv = n_kv_heads; a.a.max_len = (int)max_len;
        a.a.scale_log2e = 1.4426950408889634f / sqrtf((float)head_dim);
        a.nseg = ns;
        int64_t wgs = 0;
        for (int i = 0; i < ns; ++i) {
            const onebit_seg_t &g = segs[s0 + i];
            const int nmb = (g.n + OB_FL_BM - 1) / OB_FL_BM;
            wgs += (int64_t)((nmb + 1) / 2) * n_heads;
            if (wgs > 0x3fffffffLL) return ob_fail(ONEBIT_E_ARG, "attention_ragged: too many workgroups");
            a.wg_end[i] = (int)wgs;
            a.seg[i] = {g.row0, g.n, g.slot, g.past};
        }
        if (head_dim == 128) hipLaunchKernelGGL((ob_flash_fwd_kernel<128, true>), dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((ob_flash_fwd_kernel<64, true>), dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
        const int rc = ob_launch_status("attention_ragged");
        if (rc) return rc;
    }
    return 0;
}

// -------------------------------------------------------------------------------------------- split-KV decode attention --
static inline size_t ob_align256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t onebit_attention_decode_scratch_bytes(int64_t rows, int32_t n_heads, int32_t n_splits)
{
    if (rows <= 0 || n_heads <= 0 || n_splits <= 1) return 0;      // one split: no partials, no tickets
    const size_t rh = (size_t)rows * (size_t)n_heads;
    return ob_align256(rh * n_splits * 128 * 4) + ob_align256(rh * n_splits * 2 * 4) + ob_align256(rh * 4);
}

// `cap_rows` >= rows: the row count the scratch is laid out for (tickets [cap_rows][H] | {max, sum} [cap_rows][H][n_splits][2] |
// partial outputs [cap_rows][H][n_splits][128]): tickets FIRST, their place depends on (cap_rows, n_heads) only.
// `fu` (optional): the FUSED form's inputs (ObFdecArgs fields u_q .. ln_eps filled in; q unused).
static int ob_fdec_launch(const void *q, const void *k_cache, const void *v_cache, void *o, const void *h_next, const int32_t *row_slot,
                          const int32_t *row_pos, int64_t rows, int64_t cap_rows, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim,
                          int64_t n_slots, int64_t max_len, int32_t chunk, int32_t n_splits, void *scratch, hipStream_t stream,
                          const ObFdecArgs *fu = nullptr)
{
    ObFdecArgs a = {};
    if (fu) a = *fu;
    a.q = (const _Float16 *)q; a.k = (const _Float16 *)k_cache; a.v = (const _Float16 *)v_cache; a.o = (_Float16 *)o;
    a.h_next = (const _Float16 *)h_next; a.row_slot = row_slot; a.row_pos = row_pos;
    a.H = n_heads; a.Hkv = n_kv_heads; a.D = head_dim; a.max_len = (int)max_len; a.n_slots = (int)n_slots; a.chunk = chunk; a.nsplit = n_splits;
    a.inv_sqrt_d = 1.0f / sqrtf((float)head_dim);
    if (n_splits > 1) {
        const size_t rh = (size_t)cap_rows * (size_t)n_heads;
        char *p = (char *)scratch;
        a.counter = (int *)p; p += ob_align256(rh * 4);
        a.part_ml = (float *)p; p += ob_align256(rh * n_splits * 2 * 4);
        a.part_o = (float *)p;
    }
    const dim3 grid((unsigned)n_heads, (unsigned)rows, (unsigned)n_splits);
    // keys per thread in flight: a 128-position multiple sweeps 8 per thread (one round trip per 128 positions), else 4
    if (fu) {
        if (chunk % 128 == 0) hipLaunchKernelGGL((ob_fdec_kernel<8, true>), grid, dim3(OB_FD_THREADS), 0, stream, a);
        else hipLaunchKernelGGL((ob_fdec_kernel<4, true>), grid, dim3(OB_FD_THREADS), 0, stream, a);
    } else if (chunk % 128 == 0) hipLaunchKernelGGL((ob_fdec_kernel<8>), grid, dim3(OB_FD_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((ob_fdec_kernel<4>), grid, dim3(OB_FD_THREADS), 0, stream, a);
    return ob_launch_status("attention_decode_rows");
}

static int ob_fdec_check(const char *fn, int64_t rows, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t n_slots, int64_t max_len,
                         int32_t chunk, int32_t n_splits, const void *scratch, size_t scratch_bytes)
{
    if (rows < 0 || n_heads <= 0 || n_kv_heads <= 0 || n_slots <= 0 || max_len <= 0) return ob_fail(ONEBIT_E_ARG, "%s: bad size", fn);
    if (head_dim < 8 || head_dim % 8 != 0 || head_dim > 128 || n_heads % n_kv_heads != 0)
        return ob_fail(ONEBIT_E_SHAPE, "%s: head_dim %d (multiple of 8, <= 128), heads %d / %d", fn, head_dim, n_heads, n_kv_heads);
    if (chunk < 64 || chunk % 64 != 0 || n_splits < 1 || n_splits > 65535)
        return ob_fail(ONEBIT_E_SHAPE, "%s: chunk %d (a multiple of 64) x %d splits", fn, chunk, n_splits);
    if (rows > 65535 || max_len > 0x7fffffffLL) return ob_fail(ONEBIT_E_ARG, "%s: dimension too large", fn);
    const size_t need = onebit_attention_decode_scratch_bytes(rows, n_heads, n_splits);
    if (rows > 0 && need && (!scratch || scratch_bytes < need || !ob_aligned(scratch, 16)))
        return ob_fail(ONEBIT_E_WSPACE, "%s: needs %zu bytes of zero-initialised, 16-byte aligned scratch", fn, need);
    return 0;
}

// The FUSED form: q (and, in the last live split, k / v + the cache append) formed inside the attention launch from the pre-LayerNorm
// projection rows and the producer's tile partials -- what the decode engines call when those partials exist (one launch less per layer).
extern "C" int onebit_attention_decode_rows_fused(const void *u_q, const void *u_k, const void *u_v, const float *st_q, const float *st_k,
                                                  const float *st_v, const void *q_bias, const void *k_bias, const void *v_bias,
                                                  const void *cos, const void *sin, void *k_cache, void *v_cache, void *o, const void *h_next,
                                                  const int32_t *row_slot, const int32_t *row_pos, int64_t rows, int32_t n_heads,
                                                  int32_t n_kv_heads, int32_t head_dim, int64_t n_slots, int64_t max_len, int64_t max_pos,
                                                  int32_t chunk, int32_t n_splits, float ln_eps, void *scratch, size_t scratch_bytes, void *stream)
{
    const int rc = ob_fdec_check("attention_decode_rows_fused", rows, n_heads, n_kv_heads, head_dim, n_slots, max_len, chunk, n_splits, scratch, scratch_bytes);
    if (rc) return rc;
    if (head_dim < 16 || (head_dim & (head_dim - 1)) != 0) return ob_fail(ONEBIT_E_SHAPE, "attention_decode_rows_fused: head_dim %d must be a power of two >= 16", head_dim);
    if (((int64_t)n_heads * head_dim) % 16 != 0 || ((int64_t)n_kv_heads * head_dim) % 16 != 0 || (int64_t)n_heads * head_dim > 16384)
        return ob_fail(ONEBIT_E_SHAPE, "attention_decode_rows_fused: the tile partials need whole 16-row tiles and rows of at most 16384 elements");
    if (max_len > max_pos) return ob_fail(ONEBIT_E_SHAPE, "attention_decode_rows_fused: cache rows (%lld) beyond the rope tables (%lld)", (long long)max_len, (long long)max_pos);
    if (rows == 0) return 0;
    if (!u_q || !u_k || !u_v || !st_q || !st_k || !st_v || !cos || !sin || !k_cache || !v_cache || !o || !row_pos)
        return ob_fail(ONEBIT_E_ARG, "attention_decode_rows_fused: null pointer");
    if ((q_bias || k_bias || v_bias) && !(q_bias && k_bias && v_bias)) return ob_fail(ONEBIT_E_ARG, "attention_decode_rows_fused: some but not all biases");
    if (!ob_aligned(k_cache, 16) || !ob_aligned(v_cache, 16) || !ob_aligned(st_q, 16) || !ob_aligned(st_k, 16) || !ob_aligned(st_v, 16))
        return ob_fail(ONEBIT_E_ALIGN, "attention_decode_rows_fused: caches and tile partials must be 16-byte aligned");
    ObFdecArgs f = {};
    f.u_q = (const _Float16 *)u_q; f.u_k = (const _Float16 *)u_k; f.u_v = (const _Float16 *)u_v; f.st_q = st_q; f.st_k = st_k; f.st_v = st_v;
    f.b_q = (const _Float16 *)q_bias; f.b_k = (const _Float16 *)k_bias; f.b_v = (const _Float16 *)v_bias;
    f.cos = (const _Float16 *)cos; f.sin = (const _Float16 *)sin; f.kw = (_Float16 *)k_cache; f.vw = (_Float16 *)v_cache; f.ln_eps = ln_eps;
    return ob_fdec_launch(nullptr, k_cache, v_cache, o, h_next, row_slot, row_pos, rows, rows, n_heads, n_kv_heads, head_dim, n_slots, max_len, chunk,
                          n_splits, scratch, (hipStream_t)stream, &f);
}

extern "C" int onebit_attention_decode_rows(const void *q, const void *k_cache, const void *v_cache, void *o, const void *h_next,
                                            const int32_t *row_slot, const int32_t *row_pos, int64_t rows, int32_t n_heads,
                                            int32_t n_kv_heads, int32_t head_dim, int64_t n_slots, int64_t max_len, int32_t chunk,
                                            int32_t n_splits, void *scratch, size_t scratch_bytes, void *stream)
{
    const int rc = ob_fdec_check("attention_decode_rows", rows, n_heads, n_kv_heads, head_dim, n_slots, max_len, chunk, n_splits, scratch, scratch_bytes);
    if (rc) return rc;
    if (rows == 0) return 0;
    if (!q || !k_cache || !v_cache || !o || !row_pos) return ob_fail(ONEBIT_E_ARG, "attention_decode_rows: null pointer");
    if (!ob_aligned(q, 16) || !ob_aligned(k_cache, 16) || !ob_aligned(v_cache, 16) || !ob_aligned(o, 2))
        return ob_fail(ONEBIT_E_ALIGN, "attention_decode_rows: q / k / v must be 16-byte aligned");
    return ob_fdec_launch(q, k_cache, v_cache, o, h_next, row_slot, row_pos, rows, rows, n_heads, n_kv_heads, head_dim, n_slots, max_len, chunk,
                          n_splits, scratch, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------- the mixed step --
// Workspace: every [rows, *] activation of a step, carved from one caller buffer (256-byte aligned pieces).
struct ObMixedWs {
    _Float16 *hA, *hB, *x, *xs[3], *uq, *uk, *uv, *q, *attn, *uo, *ug, *uu, *act, *ud, *xh;
    char *gemm_ws; size_t gemm_ws_bytes;
    char *attn_scratch; size_t attn_scratch_bytes;
    float *z[4];                  // K-slice sums [rows, hidden] of o_proj / down_proj at a few hundred rows (ob_gemm3_ksplit)
    size_t total;
};
#define OB_MIXED_MAX_OUT 64          // rows per lm_head launch
#define OB_MIXED_MAX_SPLITS 64

static int ob_mixed_splits(const onebit_model_t *m, int chunk, int ctx)
{
    const int c = chunk > 0 ? chunk : 256;
    const int L = ctx > 0 ? std::min(ctx, (int)m->max_len) : (int)m->max_len;
    return std::max(1, (L + c - 1) / c);
}

// (`dec_rows` = the workspace's capacity for single-token rows -- n_slots --, NOT a step's n_dec: the attention scratch comes first
//  and is laid out for that many rows in every step, so the arrival tickets never move onto memory other steps have written)
static ObMixedWs ob_mixed_carve(const onebit_model_t *m, int64_t rows, int dec_rows, int n_splits, void *base)
{
    ObMixedWs w = {};
    const size_t H = m->hidden, I = m->intermediate, NQ = (size_t)m->n_heads * m->head_dim, NK = (size_t)m->n_kv_heads * m->head_dim;
    size_t off = 0;
    auto take = [&](size_t bytes) -> char * { char *p = base ? (char *)base + off : nullptr; off += ob_align256(bytes); return p; };
    const size_t T = (size_t)rows;
    w.attn_scratch_bytes = onebit_attention_decode_scratch_bytes(dec_rows, m->n_heads, OB_MIXED_MAX_SPLITS);
    w.attn_scratch = take(w.attn_scratch_bytes);
    (void)n_splits;
    w.hA = (_Float16 *)take(T * H * 2); w.hB = (_Float16 *)take(T * H * 2); w.x = (_Float16 *)take(T * H * 2);
    for (int i = 0; i < 3; ++i) w.xs[i] = (_Float16 *)take(T * H * 2);
    w.uq = (_Float16 *)take(T * NQ * 2); w.uk = (_Float16 *)take(T * NK * 2); w.uv = (_Float16 *)take(T * NK * 2);
    w.q = (_Float16 *)take(T * NQ * 2); w.attn = (_Float16 *)take(T * NQ * 2); w.uo = (_Float16 *)take(T * H * 2);
    w.ug = (_Float16 *)take(T * I * 2); w.uu = (_Float16 *)take(T * I * 2); w.act = (_Float16 *)take(T * I * 2);
    w.ud = (_Float16 *)take(T * H * 2); w.xh = (_Float16 *)take((size_t)OB_MIXED_MAX_OUT * H * 2);
    w.gemm_ws_bytes = T * std::max(std::max(H, I), NQ) * 2;
    w.gemm_ws = take(w.gemm_ws_bytes);
    for (int i = 0; i < 4; ++i) w.z[i] = (float *)take(T * H * 4);
    w.total = off;
    return w;
}

extern "C" size_t onebit_mixed_workspace_bytes(const onebit_model_t *m, int64_t max_rows, int32_t max_dec_rows, int32_t attn_chunk)
{
    if (!m || max_rows <= 0 || m->hidden <= 0 || m->intermediate <= 0 || m->n_heads <= 0 || m->n_kv_heads <= 0 || m->head_dim <= 0 || m->max_len <= 0)
        return 0;
    (void)attn_chunk;
    return ob_mixed_carve(m, max_rows, std::max<int32_t>(max_dec_rows, 1), OB_MIXED_MAX_SPLITS, nullptr).total;
}

extern "C" int onebit_mixed_step(const onebit_model_t *m, const onebit_mixed_state_t *st, void *stream)
{
    if (!m || !st) return ob_fail(ONEBIT_E_ARG, "mixed_step: null model/state");
    if (st->struct_size != sizeof(onebit_mixed_state_t))
        return ob_fail(ONEBIT_E_ARG, "mixed_step: state struct_size %llu != %zu (caller built against another ABI: this library is ABI %d)",
                       (unsigned long long)st->struct_size, sizeof(onebit_mixed_state_t), ONEBIT_ABI_VERSION);
    if (m->n_layers <= 0 || m->hidden <= 0 || m->n_heads <= 0 || m->n_kv_heads <= 0 || m->n_heads % m->n_kv_heads != 0 ||
        (m->head_dim != 64 && m->head_dim != 128) || m->hidden % 64 != 0 || m->intermediate % 8 != 0 || m->max_len <= 0 || m->vocab <= 0)
        return ob_fail(ONEBIT_E_SHAPE, "mixed_step: bad model dimensions (head_dim 64 or 128, hidden %% 64 == 0)");
    const int T = st->n_rows, ND = st->n_dec, NS = st->n_seg, NO = st->n_out;
    const int H = m->hidden, I = m->intermediate, D = m->head_dim, NQ = m->n_heads * D, NK = m->n_kv_heads * D;
    if (NQ != H) return ob_fail(ONEBIT_E_SHAPE, "mixed_step: n_heads * head_dim != hidden");
    if (T < 0 || ND < 0 || NS < 0 || NO < 0 || ND > T || st->n_slots <= 0) return ob_fail(ONEBIT_E_ARG, "mixed_step: bad row counts");
    if (T == 0) return 0;
    if (!m->layers || !m->embed || !m->final_norm_w || !m->rope_cos || !m->rope_sin || !st->tokens || !st->row_slot || !st->row_pos ||
        (NS > 0 && !st->segs) || (NO > 0 && (!st->out_rows || !st->next_tokens || !m->lm_head || !st->part_val || !st->part_idx)) || !st->workspace)
        return ob_fail(ONEBIT_E_ARG, "mixed_step: null pointer");
    // the prompt chunks' rows follow the decode rows and tile [n_dec, n_rows) without gaps, in order
    int64_t r = ND;
    for (int i = 0; i < NS; ++i) {
        const onebit_seg_t &g = st->segs[i];
        if (g.row0 != r || g.n < 1) return ob_fail(ONEBIT_E_SHAPE, "mixed_step: segment %d starts at row %d (expected %lld) with %d rows", i, g.row0, (long long)r, g.n);
        r += g.n;
    }
    if (r != T) return ob_fail(ONEBIT_E_SHAPE, "mixed_step: %d decode rows + the segments' rows = %lld, n_rows = %d", ND, (long long)r, T);
    const int chunk = st->attn_chunk > 0 ? st->attn_chunk : 256;
    int nsplit = ob_mixed_splits(m, chunk, st->dec_ctx);
    if (nsplit > OB_MIXED_MAX_SPLITS) return ob_fail(ONEBIT_E_SHAPE, "mixed_step: %d attention splits (attn_chunk %d too small for this context)", nsplit, chunk);
    if (!ob_aligned(st->workspace, 256)) return ob_fail(ONEBIT_E_ALIGN, "mixed_step: workspace must be 256-byte aligned");
    const ObMixedWs w = ob_mixed_carve(m, T, st->n_slots, nsplit, st->workspace);
    if (st->workspace_bytes < w.total) return ob_fail(ONEBIT_E_WSPACE, "mixed_step: workspace %zu < %zu bytes", st->workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // one 1-bit projection over all rows: u = fp16(fp16(W . a) * g), LayerNorm left to the consumer (ONEBIT_FLAG_SKIP_LN)
    auto pres_ok = [&](const onebit_proj_t &p) -> bool {
        return onebit_linear_prescaled_ok(T, p.K, p.N, ONEBIT_F16) && p.ldw_bytes % 16 == 0 && ob_aligned(p.weight, 16) &&
               p.N * p.ldw_bytes < ((int64_t)1 << 32);
    };
    auto check = [&](const onebit_proj_t &p, int64_t K, int64_t N, const char *name, int l) -> int {
        if (!p.weight || !p.input_factor || !p.weight_scale || p.K != K || p.N != N || p.K % 32 != 0 || p.ldw_bytes < p.K / 8)
            return ob_fail(ONEBIT_E_SHAPE, "mixed_step: projection %s of layer %d has an unexpected shape", name, l);
        if (!ob_aligned(p.input_factor, 16) || !ob_aligned(p.weight_scale, 16)) return ob_fail(ONEBIT_E_ALIGN, "mixed_step: %s scales of layer %d must be 16-byte aligned", name, l);
        return 0;
    };
    auto gemm = [&](const onebit_proj_t &p, const void *rows, bool prescaled, void *u) -> int {
        return onebit_linear_forward(p.weight, p.ldw_bytes, rows, p.input_factor, p.weight_scale, nullptr, u, nullptr, w.gemm_ws, w.gemm_ws_bytes,
                                     T, p.K, p.N, ONEBIT_F16, m->ln_eps, ONEBIT_FLAG_SKIP_LN | (prescaled ? ONEBIT_FLAG_PRESCALED : 0u), s);
    };
    // 65 .. sk_rows rows: too few for the LDS-DMA GEMM's 256 x 128 tiles to fill the chip even grouped, and the round-1 kernels that took
    // such calls run a 95-row step at 13B widths in 22.9 ms (a 543-row step: 20.5).  The LDS-DMA skinny GEMM in balanced passes of <= 64 rows
    // re-reads the packed rows once per pass (40 MB per layer at 13B) and is still 1.5 - 2.7x faster: 95 rows 22.9 -> 8.6 ms, 159 rows 23.2 -> 11.9,
    // 231 rows 21.8 -> 14.4, 287 rows 18.7 (grouped LDS-DMA GEMM) -> 17.6; 351 rows 18.9 vs 19.1: the grouped GEMM takes over (tools/midt_probe.py).
    static const int sk_rows = getenv("OB_MIXED_SK_ROWS") ? atoi(getenv("OB_MIXED_SK_ROWS")) : 320;
    bool sk_pass = T > 64 && T <= sk_rows;
    for (int l = 0; sk_pass && l < m->n_layers; ++l) {
        const onebit_layer_t &L = m->layers[l];
        sk_pass = ob_sk3_proj_ok(L.q) && ob_sk3_proj_ok(L.k) && ob_sk3_proj_ok(L.v) && ob_sk3_proj_ok(L.o) && ob_sk3_proj_ok(L.gate) &&
                  ob_sk3_proj_ok(L.up) && ob_sk3_proj_ok(L.down);
    }
    auto sk_passes = [&](const onebit_proj_t *const *ps, void *const *us, const _Float16 *const *as, int np) -> int {
        const int nblk = (T + 63) / 64, rows = (T + nblk - 1) / nblk;
        for (int r0 = 0; r0 < T; r0 += rows) {
            const int n = std::min(rows, T - r0);
            void *ub[3] = {nullptr, nullptr, nullptr};
            const void *ab[3] = {nullptr, nullptr, nullptr};
            for (int i = 0; i < np; ++i) { ub[i] = (_Float16 *)us[i] + (size_t)r0 * ps[i]->N; ab[i] = as[i] + (size_t)r0 * ps[i]->K; }
            const int rc2 = ob_sk3_multi(ps, ub, ab, np, n, s);
            if (rc2) return rc2;
        }
        return 0;
    };
    // up to three projections sharing their input: ONE skinny launch when T <= 64, else one GEMM each
    auto gemm_group = [&](const onebit_proj_t *const *ps, void *const *us, const _Float16 *const *as, int np, bool prescaled) -> int {
        if (prescaled && T >= 2 && T <= 64) {
            const void *av[3] = {as[0], as[np > 1 ? 1 : 0], as[np > 2 ? 2 : 0]};
            if (ob_sk3_multi(ps, us, av, np, T, s) == 0) return 0;
        }
        if (prescaled && T >= 192) {             // one grouped launch of the LDS-DMA GEMM: no partly filled last round per projection
            const void *av[3] = {as[0], as[np > 1 ? 1 : 0], as[np > 2 ? 2 : 0]};
            // TAIL: when the last token tile (<= 128 rows) is what pushes the launch into one more round of workgroups (13B gate|up at 543
            // rows: 540 tiles on 512 slots; a round costs ~100 us whatever its fill), the rows of that tile go through the skinny GEMM
            // instead (one or two passes, ~30 us each) and the grouped launch covers whole rounds
            static const int tail_env = getenv("OB_MIXED_TAIL") ? atoi(getenv("OB_MIXED_TAIL")) : 128;      // A/B: 0 = off; else the most tail rows
            const int nbt = (T + 127) / 128, Tm = 128 * (nbt - 1), tail = T - Tm;
            if (tail_env > 0 && tail >= 2 && tail <= tail_env && Tm >= 192 && ob_gemm3_group_ok(ps, np, Tm)) {
                const int64_t slots = ob_gemm3_slots();
                const int64_t r_all = (ob_gemm3_group_tiles(ps, np, T) + slots - 1) / slots, r_main = (ob_gemm3_group_tiles(ps, np, Tm) + slots - 1) / slots;
                bool sk_ok = r_main < r_all;
                for (int i = 0; sk_ok && i < np; ++i) sk_ok = ob_sk3_proj_ok(*ps[i]);
                if (sk_ok && ob_gemm3_grouped(ps, us, av, np, Tm, s) == 0) {
                    const int nblk = (tail + 63) / 64, rows = (tail + nblk - 1) / nblk;
                    for (int r0 = Tm; r0 < T; r0 += rows) {
                        const int n = std::min(rows, T - r0);
                        void *ub[3] = {nullptr, nullptr, nullptr};
                        const void *ab[3] = {nullptr, nullptr, nullptr};
                        for (int i = 0; i < np; ++i) { ub[i] = (_Float16 *)us[i] + (size_t)r0 * ps[i]->N; ab[i] = as[i] + (size_t)r0 * ps[i]->K; }
                        const int rc2 = ob_sk3_multi(ps, ub, ab, np, n, s);           // (n >= 2: tail >= 2, blocks balanced)
                        if (rc2) return rc2;
                    }
                    return 0;
                }
            }
            if (ob_gemm3_grouped(ps, us, av, np, T, s) == 0) return 0;
        }
        if (prescaled && sk_pass) return sk_passes(ps, us, as, np);
        for (int i = 0; i < np; ++i) {
            const int rc2 = gemm(*ps[i], as[i], prescaled, us[i]);
            if (rc2) return rc2;
        }
        return 0;
    };
    int ks_prev = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        const onebit_layer_t &L = m->layers[l];
        if (!L.input_layernorm_w || !L.post_attention_layernorm_w || !L.k_cache || !L.v_cache)
            return ob_fail(ONEBIT_E_ARG, "mixed_step: null pointer in layer %d", l);
        if ((L.q_bias || L.k_bias || L.v_bias) && !(L.q_bias && L.k_bias && L.v_bias))
            return ob_fail(ONEBIT_E_ARG, "mixed_step: layer %d has some but not all of q_bias / k_bias / v_bias", l);
        if ((rc = check(L.q, H, NQ, "q", l)) || (rc = check(L.k, H, NK, "k", l)) || (rc = check(L.v, H, NK, "v", l)) || (rc = check(L.o, NQ, H, "o", l)) ||
            (rc = check(L.gate, H, I, "gate", l)) || (rc = check(L.up, H, I, "up", l)) || (rc = check(L.down, I, H, "down", l)))
            return rc;
        // pre-scaled rows for a group when every member takes them on its own -- or when the GROUP as one launch fills the chip although a
        // member alone would not (543 rows at 13B widths: 100 tiles per attention projection, 300 for q|k|v: the grouped LDS-DMA GEMM
        // instead of three launches of the 128 x 128 kernel)
        const onebit_proj_t *g_qkv[3] = {&L.q, &L.k, &L.v}, *g_gu[3] = {&L.gate, &L.up, nullptr};
        const bool grp_qkv = ob_gemm3_group_ok(g_qkv, 3, T), grp_gu = ob_gemm3_group_ok(g_gu, 2, T);
        // o_proj / down_proj (hidden-width outputs: half the tiles of a q|k|v group) that alone do not fill the chip run as two K-slices of
        // the same LDS-DMA GEMM; the row kernel that consumes them adds the slices (ks_prev: the previous layer's down_proj went that way)
        // (0 where the projection alone fills the chip; up to 128 rows two skinny passes are faster: 95 rows 8.6 vs 9.0 ms, 131 rows 11.7 vs 10.9)
        const bool ks_use = !(sk_pass && T <= 128);
        const int ks_o = ks_use ? ob_gemm3_ksplit_n(L.o, T) : 0, ks_down = ks_use ? ob_gemm3_ksplit_n(L.down, T) : 0;
        const bool pres_qkv = (pres_ok(L.q) && pres_ok(L.k) && pres_ok(L.v)) || grp_qkv || sk_pass, pres_o = pres_ok(L.o) || ks_o > 0 || sk_pass;
        const bool pres_gu = (pres_ok(L.gate) && pres_ok(L.up)) || grp_gu || sk_pass, pres_down = pres_ok(L.down) || ks_down > 0 || sk_pass;
        // 1. residual (+ LayerNorm of the previous down_proj) + input RMSNorm -> x, or the three consumers' pre-scaled rows
        ObRowsNormCall n1 = {};
        if (l == 0) { n1.embed = m->embed; n1.tokens = st->tokens; }
        else if (ks_prev) { n1.hres_in = w.hA; n1.z0 = w.z[0]; n1.z1 = w.z[1]; n1.z2 = ks_prev > 2 ? w.z[2] : nullptr; n1.z3 = ks_prev > 3 ? w.z[3] : nullptr; n1.g_prev = m->layers[l - 1].down.weight_scale; }
        else { n1.hres_in = w.hA; n1.u_prev = w.ud; }
        n1.rms_w = L.input_layernorm_w; n1.hres_out = w.hB; n1.T = T; n1.H = H; n1.rms_eps = m->rms_eps; n1.ln_eps = m->ln_eps;
        if (pres_qkv) {
            n1.n_scaled = 3;
            n1.h_next[0] = L.q.input_factor; n1.h_next[1] = L.k.input_factor; n1.h_next[2] = L.v.input_factor;
            n1.x_scaled[0] = w.xs[0]; n1.x_scaled[1] = w.xs[1]; n1.x_scaled[2] = w.xs[2];
        } else n1.x = w.x;
        if ((rc = ob_rows_norm(n1, s))) return rc;
        // 2. q | k | v
        {
            const onebit_proj_t *ps[3] = {&L.q, &L.k, &L.v};
            void *us[3] = {w.uq, w.uk, w.uv};
            const _Float16 *as[3] = {pres_qkv ? w.xs[0] : w.x, pres_qkv ? w.xs[1] : w.x, pres_qkv ? w.xs[2] : w.x};
            if ((rc = gemm_group(ps, us, as, 3, pres_qkv))) return rc;
        }
        // 3. LayerNorm(q, k, v) + RoPE + cache append, every row at its own (slot, position)
        if ((rc = onebit_rows_qkv_rope_ragged(w.uq, w.uk, w.uv, m->rope_cos, m->rope_sin, st->row_slot, st->row_pos, w.q, L.k_cache, L.v_cache,
                                              L.q_bias, L.k_bias, L.v_bias, T, m->n_heads, m->n_kv_heads, D, st->n_slots, m->max_len, m->max_len, m->ln_eps, s)))
            return rc;
        // 4. attention: prompt chunks on the MFMA flash kernel, single-token rows on the split-KV decode kernel
        const void *h_o = pres_o ? L.o.input_factor : nullptr;
        if (NS > 0 && (rc = onebit_attention_ragged(w.q, L.k_cache, L.v_cache, w.attn, h_o, st->segs, NS, m->n_heads, m->n_kv_heads, D,
                                                    st->n_slots, m->max_len, s)))
            return rc;
        if (ND > 0 && (rc = ob_fdec_launch(w.q, L.k_cache, L.v_cache, w.attn, h_o, st->row_slot, st->row_pos, ND, st->n_slots /* scratch rows */,
                                           m->n_heads, m->n_kv_heads, D, st->n_slots, m->max_len, chunk, nsplit, w.attn_scratch, s)))
            return rc;
        // 5. o_proj
        if (ks_o) { if ((rc = ob_gemm3_ksplit(L.o, w.attn, w.z, ks_o, T, s))) return rc; }
        else if (sk_pass) { const onebit_proj_t *p1[3] = {&L.o, nullptr, nullptr}; void *u1[3] = {w.uo, nullptr, nullptr}; const _Float16 *a1[3] = {w.attn, nullptr, nullptr};
                            if ((rc = sk_passes(p1, u1, a1, 1))) return rc; }
        else if ((rc = gemm(L.o, w.attn, pres_o, w.uo))) return rc;
        // 6. residual + LayerNorm(u_o) (+ o bias) + post-attention RMSNorm
        ObRowsNormCall n2 = {};
        n2.hres_in = w.hB; n2.bias_prev = L.o_bias;
        if (ks_o) { n2.z0 = w.z[0]; n2.z1 = w.z[1]; n2.z2 = ks_o > 2 ? w.z[2] : nullptr; n2.z3 = ks_o > 3 ? w.z[3] : nullptr; n2.g_prev = L.o.weight_scale; } else n2.u_prev = w.uo;
        n2.rms_w = L.post_attention_layernorm_w; n2.hres_out = w.hA;
        n2.T = T; n2.H = H; n2.rms_eps = m->rms_eps; n2.ln_eps = m->ln_eps;
        if (pres_gu) {
            n2.n_scaled = 2;
            n2.h_next[0] = L.gate.input_factor; n2.h_next[1] = L.up.input_factor; n2.x_scaled[0] = w.xs[0]; n2.x_scaled[1] = w.xs[1];
        } else n2.x = w.x;
        if ((rc = ob_rows_norm(n2, s))) return rc;
        // 7. gate | up, 8. SiLU(LayerNorm(gate)) * LayerNorm(up), 9. down
        {
            const onebit_proj_t *ps[3] = {&L.gate, &L.up, nullptr};
            void *us[3] = {w.ug, w.uu, nullptr};
            const _Float16 *as[3] = {pres_gu ? w.xs[0] : w.x, pres_gu ? w.xs[1] : w.x, nullptr};
            if ((rc = gemm_group(ps, us, as, 2, pres_gu))) return rc;
        }
        if ((rc = onebit_rows_swiglu(w.ug, w.uu, pres_down ? L.down.input_factor : nullptr, w.act, T, I, m->ln_eps, s))) return rc;
        if (ks_down) { if ((rc = ob_gemm3_ksplit(L.down, w.act, w.z, ks_down, T, s))) return rc; }
        else if (sk_pass) { const onebit_proj_t *p1[3] = {&L.down, nullptr, nullptr}; void *u1[3] = {w.ud, nullptr, nullptr}; const _Float16 *a1[3] = {w.act, nullptr, nullptr};
                            if ((rc = sk_passes(p1, u1, a1, 1))) return rc; }
        else if ((rc = gemm(L.down, w.act, pres_down, w.ud))) return rc;
        ks_prev = ks_down;
    }
    // final norm on the rows that sample, lm_head + greedy token (<= 64 rows per launch)
    for (int o0 = 0; o0 < NO; o0 += OB_MIXED_MAX_OUT) {
        const int no = std::min(NO - o0, OB_MIXED_MAX_OUT);
        ObRowsNormCall nf = {};
        nf.hres_in = w.hA; nf.rms_w = m->final_norm_w;
        if (ks_prev) { nf.z0 = w.z[0]; nf.z1 = w.z[1]; nf.z2 = ks_prev > 2 ? w.z[2] : nullptr; nf.z3 = ks_prev > 3 ? w.z[3] : nullptr; nf.g_prev = m->layers[m->n_layers - 1].down.weight_scale; } else nf.u_prev = w.ud;
        nf.hres_out = w.hB;      // (hB rows 0 .. no - 1: scratch here)
        nf.x = w.xh; nf.rows = st->out_rows + o0; nf.T = no; nf.H = H; nf.rms_eps = m->rms_eps; nf.ln_eps = m->ln_eps;
        if ((rc = ob_rows_norm(nf, s))) return rc;
        void *lg = st->logits ? (void *)((_Float16 *)st->logits + (size_t)o0 * m->vocab) : nullptr;
        if ((rc = ob_lm_head_argmax(w.xh, m->lm_head, lg, st->part_val, st->part_idx, st->next_tokens + o0, no, H, m->vocab, s))) return rc;
    }
    return 0;
}
