/*
 * onebit.h -- C ABI of the MI355X-native OneBit 1-bit linear layer (libonebit_hip.so).
 *
 * The reference (xuyuzhuang11/OneBit) has no FFI or plugin registry for this
 * path: its "operator API" is the Python nn.Module BitLinearInf
 * (transformers/src/transformers/models/bitnet.py:71-122) plus the sign packer
 * in scripts/convert_llama_to_infer_ckpt.py:7-15.  This header is therefore the
 * drop-in boundary the replacement module (onebit_amd/bitnet.py) binds through
 * ctypes; each entry point names the reference lines it replaces.  See
 * INTEGRATION.md for the reference-side binding.
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller (torch tensors in
 *    practice).  The library never allocates, frees or retains them.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *    only enqueue work; they never synchronise and are graph-capture safe.
 *  - `dtype` selects the floating type of x / h / g / bias / y: ONEBIT_F16 or
 *    ONEBIT_F32 (the reference computes in the dtype of weight_scale,
 *    bitnet.py:99).  Packed weights are always the reference's int8 [N, K/8]
 *    tensor (bitnet.py:78), LSB-first, bit 1 = -1; rows may be strided
 *    (`ldw_bytes`) so a K-shard can alias a column slice of the full matrix.
 *  - Return value: 0 = ok; negative = argument error (ONEBIT_E_*); positive =
 *    hipError_t from a launch.  onebit_last_error() returns a thread-local
 *    message for the last non-zero return.
 */
#ifndef ONEBIT_H
#define ONEBIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ONEBIT_ABI_VERSION 9

#define ONEBIT_F16 0
#define ONEBIT_F32 1

#define ONEBIT_E_ARG      (-1)   /* null pointer / negative size                        */
#define ONEBIT_E_SHAPE    (-2)   /* K % 8 != 0                                           */
#define ONEBIT_E_ALIGN    (-3)   /* pointer or row pitch not aligned as documented      */
#define ONEBIT_E_DTYPE    (-4)   /* unknown dtype                                       */
#define ONEBIT_E_WSPACE   (-5)   /* workspace too small                                 */
#define ONEBIT_E_FLAG     (-6)   /* unknown flag                                        */

/* flags for onebit_linear_forward */
#define ONEBIT_FLAG_SKIP_LN   1u   /* y receives u = (W.(h*x))*g, LayerNorm (and bias) skipped */
#define ONEBIT_FLAG_PRESCALED 4u   /* x already holds fp16(x * h) (bitnet.py:113 done by the producer of x:
                                      onebit_rows_res_ln_rms / onebit_rows_swiglu with h_next); h is not read.
                                      Only where onebit_linear_prescaled_ok(T, K, N, dtype) returns 1.        */

int onebit_abi_version(void);
const char *onebit_last_error(void);

/* ---- packing -------------------------------------------------------------
 * onebit_pack_signs: packed = fp16_to_int8(sign(w))
 *   replaces scripts/convert_llama_to_infer_ckpt.py:7-15 (fp16_to_int8) applied
 *   as in :29-32; bit = (w < 0), so w == 0 and NaN pack as +1.
 *   w [N,K] (dtype), packed int8 [N,K/8].  K % 8 == 0.
 * onebit_unpack_signs: dense = int8_to_fp16(packed)
 *   replaces bitnet.py:98-110.  out [N,K] (dtype) of +1 / -1.
 */
int onebit_pack_signs(const void *w, int dtype, void *packed, int64_t N, int64_t K, void *stream);
/* onebit_fp16_to_int8: the reference function itself (convert_llama_to_infer_ckpt.py:7-15) on an arbitrary
 *   tensor s [N,K] (dtype): v = (0 - s + 1) / 2 evaluated in dtype, truncated to uint8 (:10), byte j =
 *   sum_i v[8j+i] * 2^i mod 256 (:12-13).  On sign values (+1 / -1 / 0) identical to onebit_pack_signs;
 *   elsewhere it reproduces the reference's arithmetic (its asserts :8-9 are commented out): s = -0.5 packs as
 *   +1, s = -3 sets the NEXT bit.  Domain: s <= 1 (v >= 0; for s > 1 torch's float -> uint8 conversion is
 *   implementation-defined; here such elements contribute 0, like NaN).  s 16-byte aligned.               */
int onebit_fp16_to_int8(const void *s, int dtype, void *packed, int64_t N, int64_t K, void *stream);
int onebit_unpack_signs(const void *packed, void *out, int dtype, int64_t N, int64_t K, void *stream);

/* ---- forward (bitnet.py:112-122) -------------------------------------------
 * y[T,N] = LayerNorm_N( g * ( W(+-1) . (h * x[T,K]) ) ) (+ bias)
 *   dtype F16: the reference's fp16 rounding points are reproduced
 *     (a = fp16(x*h) :113; z = fp16(sum, fp32 accumulate) :115; u = fp16(z*g)
 *     :116; LayerNorm statistics in fp32, eps, biased variance :118; bias :119).
 *   dtype F32: everything in fp32.
 *   K % 8 == 0 (the reference's packing constraint).  The MFMA kernels take
 *   K % 32 == 0 with 4-byte aligned packed rows (every LLaMA shape; 16-byte
 *   aligned rows get the widest loads); other shapes run a generic kernel.
 *   x, y rows contiguous; x/h/y 16-byte aligned.  u_or_null, if given,
 *   receives the pre-LayerNorm u [T,N].  workspace: caller-owned, 16-byte
 *   aligned scratch of onebit_linear_workspace_bytes() bytes: required only
 *   for K % 32 != 0 (fp32 z); for large prefill calls it holds the pre-scaled
 *   activations fp16(x*h) of the LDS-DMA GEMM -- without it (NULL / smaller)
 *   the register-staged GEMM runs instead, same results.
 */
size_t onebit_linear_workspace_bytes(int64_t T, int64_t K, int64_t N, int dtype);
int onebit_linear_forward(const void *packed, int64_t ldw_bytes, const void *x, const void *h,
                          const void *g, const void *bias_or_null, void *y, void *u_or_null,
                          void *workspace, size_t workspace_bytes, int64_t T, int64_t K,
                          int64_t N, int dtype, float ln_eps, unsigned flags, void *stream);

/* ---- split entry points (K-sharded multi-GPU path) -----
 * onebit_matmul_partial: zp[T,N] (fp32) = W[:, Kslice] . (h[Kslice] * x[:, Kslice])
 *   No rounding of the sum, no g: partial sums to be all-reduced over ranks.
 *   x is [T, K] with row pitch ldx elements (so a K-slice of a wider activation
 *   can be passed in place); dtype of x/h as above.
 * onebit_scale_layernorm: y = LayerNorm(g * round(z)) (+bias) from fp32 z[T,N]
 *   (the epilogue of bitnet.py:115-120 applied after the all-reduce).
 */
int onebit_matmul_partial(const void *packed, int64_t ldw_bytes, const void *x, int64_t ldx,
                          const void *h, float *zp, int64_t T, int64_t K, int64_t N, int dtype,
                          void *stream);
/* The same with a caller workspace: onebit_linear_workspace_bytes(T, K, N, dtype) bytes let large
 * calls run the LDS-DMA GEMM on the pre-scaled K slice (a smaller or NULL workspace is accepted). */
int onebit_matmul_partial_ws(const void *packed, int64_t ldw_bytes, const void *x, int64_t ldx,
                             const void *h, float *zp, void *workspace, size_t workspace_bytes,
                             int64_t T, int64_t K, int64_t N, int dtype, void *stream);
int onebit_scale_layernorm(const float *z, const void *g, const void *bias_or_null, void *y,
                           void *u_or_null, int64_t T, int64_t N, int dtype, float ln_eps,
                           unsigned flags, void *stream);

/* ---- N-sharded (output rows split) LayerNorm, SURVEY.md section 8e "alternatives" -----------
 * A rank that owns columns [n0, n1) of the output holds u[T, n] (n = n1 - n0, from
 * onebit_linear_forward with ONEBIT_FLAG_SKIP_LN on its row slice of the packed matrix).  The
 * LayerNorm of bitnet.py:118 needs statistics of the COMPLETE row, so it is split in two:
 * onebit_row_stats:      stats[t] = { mean of u[t, :], sum of squared deviations from that mean }
 *                        of the rank's n columns (fp32, exact two-pass); ranks combine them with
 *                        the parallel-variance formula (host side, [T, 2] floats per rank).
 * onebit_normalize_rows: y[t, j] = (u[t, j] - mean[t]) * rstd[t] (+ bias[j]) with the combined
 *                        row statistics; dtype rounding as in the fused LayerNorm.
 * u, y: [T, n] contiguous, fp16 or fp32 (dtype); stats, mean, rstd: fp32 device arrays.
 */
int onebit_row_stats(const void *u, float *stats, int64_t T, int64_t n, int dtype, void *stream);
/* ABI 7: the same statistics WITHOUT reading u again.  onebit_linear_forward with ONEBIT_FLAG_SKIP_LN | ONEBIT_FLAG_TILE_STATS
 * (fp16; shapes for which onebit_linear_tile_stats_ok() is 1: the call takes the LDS-DMA GEMM -- pass its workspace or
 * ONEBIT_FLAG_PRESCALED -- and N % 64 == 0) writes, from the GEMM's epilogue, per token and 64-row block the pair
 * {sum, sum of squared deviations from the block mean} of the fp16 outputs into `u_or_null`, which in that call is NOT a
 * second output but fp32 [T, N / 64, 2] (8-byte aligned).  onebit_tile_stats_combine reduces them to onebit_row_stats'
 * format: stats[t] = {mean, sum of squared deviations} of the n = N columns.  T * N / 32 bytes instead of T * N * 2. */
#define ONEBIT_FLAG_TILE_STATS 8u
int onebit_linear_tile_stats_ok(int64_t T, int64_t K, int64_t N, int dtype);
int onebit_tile_stats_combine(const float *tile_stats, float *stats, int64_t T, int64_t N, void *stream);
int onebit_normalize_rows(const void *u, const float *mean, const float *rstd, const void *bias_or_null,
                          void *y, int64_t T, int64_t n, int dtype, void *stream);

/* ---- row-wise glue around the 1-bit GEMM (prefill and batched decode) ---------------------------
 * With ONEBIT_FLAG_SKIP_LN the linear kernels leave u = fp16(fp16(z) * g); the LayerNorm that ends
 * a BitLinearInf then fuses with what the decoder layer does next (modeling_bitllama.py:912-918,
 * 76-81, 257), one pass over the rows instead of five:
 * onebit_rows_res_ln_rms: r = hres_in + LayerNorm(u_prev);  hres_out = r;  x = RMSNorm(r) * rms_w
 *                         and, for i < n_scaled (<= 3): x_scaled[i] = fp16(x * h_next[i]) -- the input
 *                         scaling of the projections that consume x, for ONEBIT_FLAG_PRESCALED calls
 *                         (x may be NULL when n_scaled > 0)
 * onebit_rows_swiglu:     act = silu(LayerNorm(u_gate)) * LayerNorm(u_up); with h_next: act = fp16(act * h_next)
 * All tensors fp16, [T, H] / [T, I] contiguous; H, I % 8 == 0 and <= 16384.
 */
int onebit_rows_res_ln_rms(const void *hres_in, const void *u_prev, const void *rms_w, void *hres_out,
                           void *x_or_null, const void *const *h_next, void *const *x_scaled, int32_t n_scaled,
                           int64_t T, int64_t H, float rms_eps, float ln_eps, void *stream);
int onebit_rows_swiglu(const void *u_gate, const void *u_up, const void *h_next_or_null, void *act, int64_t T,
                       int64_t I, float ln_eps, void *stream);
/* ABI 9: onebit_rows_res_ln_rms for a producer WITH a bias (o_proj of a checkpoint with config.attention_bias):
 * r = hres_in + fp16(LayerNorm(u_prev) + bias_prev) (bitnet.py:119-120, then modeling_bitllama.py:912).  bias_prev fp16 [H] or NULL. */
int onebit_rows_res_ln_rms_bias(const void *hres_in, const void *u_prev, const void *bias_prev_or_null, const void *rms_w, void *hres_out,
                                void *x_or_null, const void *const *h_next, void *const *x_scaled, int32_t n_scaled,
                                int64_t T, int64_t H, float rms_eps, float ln_eps, void *stream);
/* 1 when a forward call of this shape may pass ONEBIT_FLAG_PRESCALED: fp16 calls that take the LDS-DMA prefill GEMM
 * (large T) or the LDS-DMA skinny GEMM (2 <= T <= 64, K % 128 == 0, K >= 512); the call itself also needs 16-byte
 * aligned packed rows / activations.  Every other kernel multiplies by input_factor on the way in and refuses the flag. */
int onebit_linear_prescaled_ok(int64_t T, int64_t K, int64_t N, int dtype);
/* The same with the LayerNorm statistics GIVEN (row_stats [T, 4] fp32: {mean, rstd} of the complete gate row, then of
 * the complete up row; 16-byte aligned): for a tensor-parallel rank whose u_gate / u_up hold only its column slice
 * of the rows (onebit_row_stats per rank -> all-gather -> parallel-variance combine).  row_stats NULL = onebit_rows_swiglu. */
int onebit_rows_swiglu_stats(const void *u_gate, const void *u_up, const void *h_next_or_null, const float *row_stats,
                             void *act, int64_t T, int64_t I, float ln_eps, void *stream);

/* Prefill glue between the q|k|v projections (called with ONEBIT_FLAG_SKIP_LN) and attention, fp16:
 * LayerNorm of the three rows (bitnet.py:118), RoPE on q and k (modeling_bitllama.py:175-181, every op
 * rounded to fp16) and the head transpose (:526-528) in one pass over T = B * S token rows.
 * q -> [B, n_heads, S, head_dim]; k, v -> cache rows [b][kv head][past_len + s][head_dim] of caches
 * laid out [slots >= B][n_kv_heads][max_len][head_dim]; cos / sin are [max_pos, head_dim];
 * head_dim a power of two >= 16.
 * ONEBIT_FLAG_Q_TOKEN_MAJOR: q stays [B, S, n_heads, head_dim] (a caller whose attention kernel takes
 * strided views then gets its output in token-major rows, ready for o_proj without a transpose copy).  */
#define ONEBIT_FLAG_Q_TOKEN_MAJOR 0x2u
int onebit_rows_qkv_rope(const void *u_q, const void *u_k, const void *u_v, const void *cos, const void *sin,
                         void *q, void *k_cache, void *v_cache, int64_t B, int64_t S, int32_t n_heads,
                         int32_t n_kv_heads, int32_t head_dim, int64_t past_len, int64_t max_len, int64_t max_pos,
                         float ln_eps, unsigned flags, void *stream);

/* The same with the LayerNorm statistics GIVEN (row_stats [T, 6] fp32: {mean, rstd} of the complete q, k and v rows):
 * tensor-parallel ranks pass the rows of their own heads (n_heads / n_kv_heads = local counts) and statistics combined
 * across ranks.  row_stats NULL = onebit_rows_qkv_rope. */
int onebit_rows_qkv_rope_stats(const void *u_q, const void *u_k, const void *u_v, const void *cos, const void *sin,
                               const float *row_stats, void *q, void *k_cache, void *v_cache, int64_t B, int64_t S,
                               int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t past_len, int64_t max_len,
                               int64_t max_pos, float ln_eps, unsigned flags, void *stream);

/* Causal prefill attention on the rows onebit_rows_qkv_rope wrote (modeling_bitllama.py:546-563: q k^T / sqrt(D), causal
 * mask, fp32 softmax, . v), flash style on MFMA: no [S, S] tensor in HBM.  fp16.
 *   q [B, S, n_heads, D] token-major (ONEBIT_FLAG_Q_TOKEN_MAJOR), k / v cache rows [B][n_kv_heads][max_len][D] of which
 *   positions 0 .. past_len + S - 1 are valid (query s sits at position past_len + s), o [B, S, n_heads, D] token-major =
 *   the rows o_proj consumes.  h_next (optional, [n_heads * D]): o <- fp16(o * h_next), o_proj's input scaling
 *   (bitnet.py:113) so that it can be called with ONEBIT_FLAG_PRESCALED.  head_dim 64 or 128.
 * Differs from the reference's eager op order as its own flash-attention switch does (LlamaFlashAttention2,
 * modeling_bitllama.py:588): scores and probabilities are not rounded to fp16 tensors on the way. */
int onebit_attention_prefill(const void *q, const void *k_cache, const void *v_cache, void *o, const void *h_next_or_null,
                             int64_t B, int64_t S, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim,
                             int64_t past_len, int64_t max_len, void *stream);

/* ---- whole-token greedy decode, batch 1 ---------------------
 * One call enqueues every kernel of one decoded token of the reference's
 * BitLlamaForCausalLMInf (modeling_bitllama.py:1512; decoder layer :856-928, attention
 * :487-585, MLP :234-259, RMSNorm :67-81, greedy argmax generation/utils.py:2540):
 * 5 launches per decoder layer + lm_head + argmax.  The token id and the position live in
 * device memory (state->token, state->pos) and are advanced by the last kernel, so the
 * same call sequence can be captured once in a HIP graph and replayed per token.
 * All tensors fp16 except packed weights (int8) and the fp32/int32 scratch noted below.
 * Requires in_features % 32 == 0 for every projection and head_dim % 8 == 0.
 */
typedef struct onebit_proj {
    const void *weight;         /* int8  [N, ldw_bytes]  packed signs                     */
    const void *input_factor;   /* fp16  [K]                                              */
    const void *weight_scale;   /* fp16  [N]                                              */
    int64_t N, K, ldw_bytes;
} onebit_proj_t;

typedef struct onebit_layer {
    onebit_proj_t q, k, v, o, gate, up, down;
    const void *input_layernorm_w;            /* fp16 [hidden] */
    const void *post_attention_layernorm_w;   /* fp16 [hidden] */
    void *k_cache, *v_cache;                  /* fp16 [n_kv_heads, max_len, head_dim] */
    /* ABI 8: projection biases of a checkpoint with config.attention_bias (modeling_bitllama.py:451-454; added after the
     * projection's LayerNorm, bitnet.py:119-120), fp16 [N] each, or NULL.  q / k / v: all three or none.             */
    const void *q_bias, *k_bias, *v_bias, *o_bias;
} onebit_layer_t;

typedef struct onebit_model {
    int32_t n_layers, hidden, intermediate, n_heads, n_kv_heads, head_dim, vocab, max_len;
    float rms_eps, ln_eps;
    const onebit_layer_t *layers;             /* HOST array of n_layers entries */
    const void *embed;                        /* fp16 [vocab, hidden]   */
    const void *final_norm_w;                 /* fp16 [hidden]          */
    const void *lm_head;                      /* fp16 [vocab, hidden]   */
    const void *rope_cos, *rope_sin;          /* fp16 [max_pos, head_dim] (fp32 tables cast to fp16, :87-113) */
} onebit_model_t;

typedef struct onebit_decode_state {
    /* ABI 8: sizeof(onebit_decode_state_t) as the CALLER compiled it.  The library refuses a state of another size
     * (ONEBIT_E_ARG) instead of reading appended fields past the end of a struct built against an older header.     */
    uint64_t struct_size;
    int32_t *token;             /* device: in = token to process, out = greedy next token      */
    int32_t *pos;               /* device: tokens already in the KV cache; incremented         */
    int32_t *out_tokens;        /* device [max_out] or NULL: out_tokens[pos_before] = next     */
    int32_t max_out;
    void *hres0, *hres1;        /* fp16 [hidden] residual stream ping-pong                     */
    void *u_q, *u_k, *u_v;      /* fp16 [n_heads*head_dim], [n_kv*head_dim] x2                 */
    void *attn_out, *u_o;       /* fp16 [hidden]                                               */
    void *u_gate, *u_up;        /* fp16 [intermediate]                                         */
    void *u_down;               /* fp16 [hidden]                                               */
    void *logits;               /* fp16 [vocab]                                                */
    float *part_val;            /* fp32 [1024] argmax partials                                 */
    int32_t *part_idx;          /* int32 [1024]                                                */
    /* long contexts: attention split over the positions (two launches per layer over head x split,
     * deterministic last-arriver combine).  attn_splits <= 1 or attn_scratch == NULL: one workgroup
     * per head reads the whole history (the fast form up to a few hundred tokens).                 */
    int32_t attn_splits;        /* S, 2..16                                                    */
    void *attn_scratch;         /* onebit_attn_scratch_bytes(model, S) bytes, zero-filled once */
    /* Per-tile LayerNorm partials: every GEMV launch publishes, per 16 output rows, (sum, sum of
     * squared deviations from the tile mean) of its pre-LayerNorm u; the consuming launch combines
     * them (parallel-variance formula) instead of re-reading and re-reducing the vector in each of
     * its workgroups.  fp32 [onebit_decode_stats_floats(model)], caller-owned, never read before
     * it is written (no initialisation needed).                                                  */
    float *tile_stats;
    /* ABI 7, optional: fp16 [2 * head_dim] scratch.  With it the first launch of a step copies the rotary rows
     * cos[pos], sin[pos] there and the 32 attention launches read THAT instead of chasing pos -> cos / sin
     * themselves (one dependent memory round trip less on every layer's critical path).  NULL: as before.    */
    void *rope_cur;
    /* ABI 8, optional: 64 = the single-launch attention requests only the first 64 cached positions before it knows the
     * position (32 KB per head instead of 64 KB: the launch is bound by that fetch) and streams the rest once it does.  A
     * PERFORMANCE hint for steps the host knows to be at a position < 64 -- results do not depend on it at any position.
     * 0 / 128: the 128-position window.                                                                              */
    int32_t attn_blind;
    /* ABI 9, optional: attn_chunk > 0 (a multiple of 64) with q_rows set selects the KEY-BLOCK form of the long-context attention:
     * one launch normalises / rotates q, k, v and appends to the cache (onebit_rows_qkv_rope_ragged), one launch streams the
     * context as attn_splits workgroups of attn_chunk positions per head with an in-launch last-arriver combine
     * (onebit_attention_decode_rows; attn_scratch then holds onebit_attention_decode_scratch_bytes(1, n_heads, attn_splits) bytes,
     * zero-filled once) -- K and V are each read once, no score scratch; attn_splits * attn_chunk must cover the context + 1.
     * attn_chunk == 0: the round-2 pair (scores kernel + P.V kernel, exact fp16 probabilities).                             */
    int32_t attn_chunk;
    void *q_rows;               /* fp16 [n_heads * head_dim]                                                               */
} onebit_decode_state_t;

size_t onebit_attn_scratch_bytes(const onebit_model_t *model, int32_t splits);
size_t onebit_decode_stats_floats(const onebit_model_t *model);

int onebit_decode_step(const onebit_model_t *model, const onebit_decode_state_t *state, void *stream);

/* ---- batched decode step: one new token for each of B sequences (BASELINE config 5) ----------
 * The same decoder arithmetic as onebit_decode_step, organised for weight reuse: every 1-bit
 * projection is ONE skinny-GEMM launch over the [B, K] activations of all sequences (packed
 * weights streamed once per step), the row-wise glue (residual + LayerNorm + RMSNorm, SiLU * up)
 * runs once per row, attention once per (head, sequence) on that sequence's KV-cache slot.
 * 8 launches per decoder layer (q|k|v and gate|up share one launch each) + the final norm (x_out is the
 * final-norm output [B, hidden]) + optionally the batched lm_head and the per-sequence argmax.
 * layer->k_cache / v_cache here are [B][n_kv_heads][max_len][head_dim]; pos[b] < 0 marks an idle
 * slot (its row is computed but attention and the cache append are skipped).  2 <= B <= 64.
 */
typedef struct onebit_batch_state {
    uint64_t struct_size;       /* ABI 8: sizeof(onebit_batch_state_t) as the caller compiled it (checked, see above) */
    int32_t batch;              /* B                                                            */
    const int32_t *tokens;      /* device [B]: token to process per slot                        */
    const int32_t *pos;         /* device [B]: tokens already cached per slot, < 0 = idle       */
    void *hres0, *hres1;        /* fp16 [B, hidden] residual stream ping-pong                   */
    void *x;                    /* fp16 [B, hidden] normalised activations (also the output)    */
    void *act;                  /* fp16 [B, intermediate] SiLU(gate) * up                       */
    void *u_q, *u_k, *u_v;      /* fp16 [B, n_heads*head_dim], [B, n_kv*head_dim] x2            */
    void *attn_out, *u_o;       /* fp16 [B, hidden]                                             */
    void *u_gate, *u_up;        /* fp16 [B, intermediate]                                       */
    void *u_down;               /* fp16 [B, hidden]                                             */
    /* optional: lm_head + greedy sampling inside the step (modeling_bitllama.py:1610-1611,
     * generation/utils.py:2540).  With next_tokens != NULL (and model->lm_head set, hidden % 64 == 0)
     * the fp16 lm_head matrix is streamed once for all B rows and next_tokens[b] = argmax of row
     * b's logits (first index on ties); logits, if given, receives the fp16 logits.              */
    int32_t *next_tokens;       /* device [B] out, or NULL: the caller does lm_head / sampling  */
    void *logits;               /* fp16 [B, vocab] or NULL                                      */
    float *part_val;            /* fp32 [ceil(vocab / 128) * 64] scratch (with next_tokens)     */
    int32_t *part_idx;          /* int32 [ceil(vocab / 128) * 64] scratch (with next_tokens)    */
    /* optional: fp32 [onebit_batch_stats_floats(model, B)] scratch.  With it the q|k|v GEMM publishes
     * the per-16-row-tile LayerNorm partials of its three output rows per slot and the attention
     * workgroups combine them instead of re-reducing the rows (same values up to fp32 rounding of
     * mean / rstd).  NULL: recompute per workgroup.                                                 */
    float *qkv_stats;
    /* optional: fp16 [3, B, hidden] scratch.  With it the row kernels write the pre-scaled rows fp16(x * input_factor)
     * of every consuming projection (the rounding of bitnet.py:113 done once by the producer) and all seven projections
     * of a layer take the LDS-DMA skinny GEMM (same sums, fp32 accumulation order differs).  NULL: the projections
     * scale x themselves.                                                                                          */
    void *x_scaled;
    /* ABI 7: independent chains.  0 / 1: all B rows in one launch chain.  2..4: the rows as that many groups of consecutive
     * slots, each group's layers on its own HIP stream forked from / joined to `stream` (parallel branches under graph
     * capture), one lm_head over all rows.  Per-row results do not depend on the grouping.  The side streams are created
     * on the first such call, which must not be inside a stream capture.                                            */
    int32_t chains;
    /* ABI 9: attention over KEY BLOCKS for contexts of any length.  attn_splits >= 1 with q_rows (and, for attn_splits > 1,
     * attn_scratch) set: per layer the q | k | v rows go through onebit_rows_qkv_rope_ragged (LayerNorm + RoPE + cache append, one
     * workgroup per slot) into q_rows and the attention runs as onebit_attention_decode_rows: (head, slot, split) workgroups of
     * attn_chunk positions each, combined by the last arriver -- 9 launches per layer instead of 8, no bound on max_len, and a
     * 512-token context is streamed by 2-8 workgroups per (head, slot) instead of one.  attn_splits * attn_chunk must cover the
     * longest context of the step (+ 1); a caller that knows its positions sizes it per step (or per captured graph).
     * attn_splits == 0: one workgroup per (head, slot) keeps every score in LDS (the fast form up to a few hundred positions;
     * 4 * max_len + 5.4 KB of LDS <= 64 KB).  Not combined with chains > 1.                                                  */
    int32_t attn_splits;
    int32_t attn_chunk;         /* positions per split, a multiple of 64 (0: 256)                                            */
    void *q_rows;               /* fp16 [B, n_heads * head_dim]                                                             */
    void *attn_scratch;         /* onebit_attention_decode_scratch_bytes(B, n_heads, attn_splits) bytes, zero-filled once   */
} onebit_batch_state_t;

size_t onebit_batch_stats_floats(const onebit_model_t *model, int32_t batch);
int onebit_decode_step_batched(const onebit_model_t *model, const onebit_batch_state_t *state, void *stream);

/* Test support: fill all 160 KiB of LDS on every CU with `pattern` (LDS keeps its contents between launches); used by the
 * GPU tests to show that no decode launch depends on stale zeros in the padded part of its LDS images.                  */
int onebit_debug_fill_lds(uint32_t pattern, void *stream);

/* ---- K-sharded decode step (BASELINE config 4: LLaMA-13B decode, hidden dim sharded over 2 / 4 / 8 GPUs) -------------
 * z = W+- . (h * x) is linear in K, so rank p keeps the byte-column slice W[:, K_p] of every packed
 * matrix and h[K_p], multiplies its slice and the fp32 partial sums of the ranks are added BEFORE the rounding points of
 * bitnet.py:115-118.  The exchange is the CALLER's (torch.distributed all_reduce over RCCL: this library has no
 * communicator), so the step comes in SEGMENTS; between two segments the caller all-reduces the named fp32 buffer in place:
 *
 *   for every layer l:   segment ONEBIT_KSEG_QKV      -> all_reduce(z_qkv  [n_heads*D + 2*n_kv*D])      (q | k | v in ONE call)
 *                        segment ONEBIT_KSEG_ATTN_O   -> all_reduce(z_o    [hidden])
 *                        segment ONEBIT_KSEG_GATE_UP  -> all_reduce(z_gu   [2 * intermediate])           (gate | up in ONE call)
 *                        segment ONEBIT_KSEG_DOWN     -> all_reduce(z_down [hidden])
 *   then once:           segment ONEBIT_KSEG_HEAD     (final norm, fp16 lm_head, greedy token, position += 1)
 *
 * = 4 collectives per layer instead of one per BitLinearInf call (7), every launch a native kernel (the decode GEMV in its
 * fp32-partial form on the rank's K slice, one-workgroup row kernels for the replicated glue, the decode attention kernel --
 * the consumers of a reduced sum round and scale it themselves: 8 launches per layer), capturable with the collectives in ONE
 * HIP graph.  Everything that is not a K-sliced product
 * (LayerNorm / RMSNorm / RoPE / attention over the replicated KV cache / lm_head) is computed by every rank on identical
 * inputs, hence identical results.  world = 1: no collective, the same arithmetic.
 *
 * The MODEL passed here describes the rank's slices: every onebit_proj_t has weight = first byte of the slice inside the
 * full packed matrix (or a copy), K = slice width (a multiple of 128; weight and input_factor 16-byte aligned), ldw_bytes =
 * the row pitch, input_factor = h[K_p], weight_scale (and the layer's biases) FULL; hidden / intermediate / heads are the
 * full model's.  k0_* = first column of the slice inside the full input vector of the projections with that in_features.   */
enum { ONEBIT_KSEG_QKV = 0, ONEBIT_KSEG_ATTN_O = 1, ONEBIT_KSEG_GATE_UP = 2, ONEBIT_KSEG_DOWN = 3, ONEBIT_KSEG_HEAD = 4 };
typedef struct onebit_kshard_state {
    uint64_t struct_size;       /* sizeof(onebit_kshard_state_t) as the caller compiled it (checked)                 */
    int32_t *token, *pos, *out_tokens;      /* as onebit_decode_state_t                                             */
    int32_t max_out;
    void *hres0, *hres1;        /* fp16 [hidden] residual stream ping-pong                                           */
    void *x;                    /* fp16 [hidden] normalised input of the next projections                            */
    void *u_q, *u_k, *u_v;      /* fp16 [n_heads*D], [n_kv*D] x 2: scratch (unused since the consumers read z directly) */
    void *attn_out;             /* fp16 [n_heads*D]                                                                   */
    void *u_gate, *u_up, *act;  /* fp16 [intermediate] x 3                                                           */
    void *u_down;               /* fp16 [hidden]                                                                      */
    float *z_qkv, *z_o, *z_gu, *z_down;     /* fp32 partial sums OUT of a segment, complete sums INTO the next one    */
    void *logits;               /* fp16 [vocab]                                                                       */
    float *part_val;            /* fp32 [1024] argmax partials                                                        */
    int32_t *part_idx;          /* int32 [1024]                                                                       */
    float *tile_stats;          /* fp32 [onebit_decode_stats_floats(model)]                                           */
    int32_t k0_hidden, k0_attn, k0_inter;   /* slice origin for in_features = hidden / n_heads*D / intermediate       */
    /* (appended in ABI 9) attn_chunk > 0: the attention of ONEBIT_KSEG_ATTN_O takes the key-block form of the other engines
     * (onebit_attention_decode_rows_fused: attn_splits workgroups per head, each over attn_chunk cached tokens, attn_chunk % 64 == 0,
     * attn_splits * attn_chunk >= max_len) after one launch that rounds and scales the reduced q | k | v sums into u_q / u_k / u_v
     * with their LayerNorm partials -- for contexts where one workgroup per head streaming the whole cache is the step's longest
     * launch, and for max_len beyond the one-workgroup kernel's 64 KB of LDS.  head_dim a power of two >= 16.
     * attn_scratch: onebit_attention_decode_scratch_bytes(1, n_heads, attn_splits) bytes, zeroed once by the caller.          */
    int32_t attn_chunk, attn_splits;
    void *attn_scratch;
} onebit_kshard_state_t;
int onebit_decode_step_ksharded(const onebit_model_t *model, const onebit_kshard_state_t *state, int32_t layer,
                                int32_t segment, void *stream);

/* ABI 9: up to three projections that share their token rows' count T and in_features K (q | k | v, gate | up of a decoder layer), each on
 * ITS OWN pre-scaled rows a[i] = fp16(x * input_factor_i) [T, K] (onebit_rows_res_ln_rms with h_next), in ONE launch:
 * u[i] [T, N_i] = fp16(fp16(W_i . a_i) * g_i), the pre-LayerNorm output (ONEBIT_FLAG_SKIP_LN | ONEBIT_FLAG_PRESCALED semantics of
 * onebit_linear_forward, bit-identical to it).  One launch instead of three keeps the workgroup rounds full (T = 2048, 7B: three
 * launches of 256 tiles each on 512 workgroup slots vs one of 768).  2 <= T <= 64: the LDS-DMA skinny GEMM; larger T: the LDS-DMA
 * GEMM.  Returns ONEBIT_E_SHAPE when the group is not eligible (the caller then issues one onebit_linear_forward per projection). */
int onebit_linear_group_prescaled(const onebit_proj_t *projs, void *const *u, const void *const *a, int32_t n_proj, int64_t T, void *stream);

/* ---- ragged token rows: several sequences in one call (continuous batching, BASELINE config 5; ABI 9) -----------------
 * The reference runs one rectangular batch per forward (modeling_bitllama.py:1275-1330); a continuous-batching step instead
 * concatenates the token rows of all scheduled requests: item i = rows [row0, row0 + n) of the request whose KV cache is slot
 * `slot`, its first new token at position `past` (keys 0 .. past + n - 1 of that slot are valid once the rows' keys / values
 * have been appended).  The arithmetic per row is the reference's (attention with past: :487-585). */
typedef struct onebit_seg { int32_t row0, n, slot, past; } onebit_seg_t;

/* onebit_rows_qkv_rope for ragged rows: LayerNorm of the three pre-LayerNorm rows (bitnet.py:118), RoPE at the row's OWN position
 * (modeling_bitllama.py:175-181), k / v appended to cache row [row_slot[t]][kv head][row_pos[t]], q token-major [T, n_heads, D].
 * row_slot (NULL: slot = t) / row_pos: DEVICE int32 [T]; a row whose slot / position lies outside [0, n_slots) x [0, max_len)
 * is skipped (an idle decode slot).  caches [n_slots][n_kv_heads][max_len][head_dim]; cos / sin [max_pos >= max_len, head_dim]. */
/* q_bias / k_bias / v_bias: the projections' biases of a checkpoint with config.attention_bias (modeling_bitllama.py:451-453), fp16
 * [n_heads * D] / [n_kv_heads * D] x 2, all three or all NULL: q / k / v = fp16(LayerNorm(u) + b) (bitnet.py:118-120) before RoPE. */
int onebit_rows_qkv_rope_ragged(const void *u_q, const void *u_k, const void *u_v, const void *cos, const void *sin,
                                const int32_t *row_slot, const int32_t *row_pos, void *q, void *k_cache, void *v_cache,
                                const void *q_bias, const void *k_bias, const void *v_bias,
                                int64_t T, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t n_slots,
                                int64_t max_len, int64_t max_pos, float ln_eps, void *stream);

/* onebit_attention_prefill over ragged segments: causal attention of every segment's n queries (rows row0 .. of q, token-major
 * [rows, n_heads, D]) against keys 0 .. past + n - 1 of its slot; o [rows, n_heads, D] (x h_next when given).  `segs` is a HOST
 * array (it travels in the kernel arguments, 64 segments per launch: nothing to upload, graph-capture safe).  head_dim 64 / 128. */
int onebit_attention_ragged(const void *q, const void *k_cache, const void *v_cache, void *o, const void *h_next_or_null,
                            const onebit_seg_t *segs, int32_t n_seg, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim,
                            int64_t n_slots, int64_t max_len, void *stream);

/* Decode attention for single-token rows at ANY context length (modeling_bitllama.py:546-563 with q_len = 1): row r attends to keys
 * 0 .. row_pos[r] of slot row_slot[r] (NULL: slot r; the row's own key / value already appended: onebit_rows_qkv_rope_ragged), the
 * positions split over `n_splits` workgroups of `chunk` positions each per (row, head) (flash-decoding; chunk a multiple of 64,
 * chunk * n_splits >= the longest context + 1 -- positions beyond that are NOT attended), partial results combined by the last
 * workgroup to arrive, in split order (deterministic).  q [rows, n_heads, D] post-RoPE, o [rows, n_heads * D] (x h_next).
 * scratch: onebit_attention_decode_scratch_bytes(rows, n_heads, n_splits) bytes, ZERO-FILLED ONCE by the caller (the arrival tickets
 * -- the first rows * n_heads int32 of it, whatever n_splits is -- return to zero after every launch; a scratch that is reused with
 * another `rows` or `n_heads` must be zero-filled again); n_splits == 1 needs none.  Scores are rounded as the reference's eager attention rounds them;
 * probabilities are not rounded to fp16 before the value product (as in onebit_attention_prefill). */
size_t onebit_attention_decode_scratch_bytes(int64_t rows, int32_t n_heads, int32_t n_splits);
int onebit_attention_decode_rows(const void *q, const void *k_cache, const void *v_cache, void *o, const void *h_next_or_null,
                                 const int32_t *row_slot, const int32_t *row_pos, int64_t rows, int32_t n_heads, int32_t n_kv_heads,
                                 int32_t head_dim, int64_t n_slots, int64_t max_len, int32_t chunk, int32_t n_splits, void *scratch,
                                 size_t scratch_bytes, void *stream);

/* The same with the query formed INSIDE the attention launch (no separate rope / append launch): u_q / u_k / u_v are the pre-LayerNorm
 * projection rows [rows, n_heads * D] / [rows, n_kv_heads * D] x 2, st_q / st_k / st_v the producer's per-16-row-tile LayerNorm partials
 * of those rows (ob_tile layout of onebit_decode_state_t.tile_stats / onebit_batch_state_t.qkv_stats: ceil(n / 4096) * 512 floats per
 * row); every (head, row, split) workgroup normalises and rotates its own head's query, the workgroup of the last live split forms the
 * new key / value, attends to them from LDS and -- one workgroup per kv head -- appends them to the cache.  Biases as above. */
int onebit_attention_decode_rows_fused(const void *u_q, const void *u_k, const void *u_v, const float *st_q, const float *st_k,
                                       const float *st_v, const void *q_bias, const void *k_bias, const void *v_bias,
                                       const void *cos, const void *sin, void *k_cache, void *v_cache, void *o, const void *h_next_or_null,
                                       const int32_t *row_slot, const int32_t *row_pos, int64_t rows, int32_t n_heads, int32_t n_kv_heads,
                                       int32_t head_dim, int64_t n_slots, int64_t max_len, int64_t max_pos, int32_t chunk, int32_t n_splits,
                                       float ln_eps, void *scratch, size_t scratch_bytes, void *stream);

/* ---- mixed prefill + decode step (BASELINE config 5: "mixed prefill+decode continuous batch") ---------------------------
 * ONE scheduler step on native kernels: the token rows of all scheduled items -- first the single next token of every decoding
 * request (n_dec rows), then the prompt chunks of the requests still entering (n_seg segments) -- go through every 1-bit
 * projection as ONE GEMM over all n_rows rows (packed weights streamed once per step), the row glue runs fused
 * (residual + LayerNorm + RMSNorm writing the consumers' pre-scaled rows; LayerNorm + RoPE + cache append per row at its own
 * (slot, position); SiLU(LayerNorm) * LayerNorm), attention runs per item on its own cache slot (prompt chunks: causal flash
 * attention with past on MFMA; decode rows: split-KV), and the lm_head + greedy token run on the n_out rows that sample.
 * 9-13 launches per decoder layer whatever the number of items; no per-item loop on the host.
 * model->layers[l].k_cache / v_cache: [n_slots][n_kv_heads][max_len][head_dim].  Checkpoints with config.attention_bias are
 * taken (q / k / v biases in the rope kernel, the o bias in the norm kernel that consumes u_o).                              */
typedef struct onebit_mixed_state {
    uint64_t struct_size;        /* sizeof(onebit_mixed_state_t) as the caller compiled it (checked)                          */
    int32_t n_rows;              /* token rows of the step                                                                    */
    int32_t n_dec;               /* rows 0 .. n_dec - 1: one token each of n_dec different requests                           */
    int32_t n_seg;               /* prompt chunks; their rows tile [n_dec, n_rows) in order                                   */
    int32_t n_out;               /* rows whose greedy next token is wanted                                                    */
    int32_t n_slots;             /* slots of the KV caches                                                                    */
    int32_t attn_chunk;          /* positions per split of the decode rows' attention (0: 256)                                */
    int32_t dec_ctx;             /* host-known bound on row_pos + 1 over the decode rows (sizes the split grid); 0: max_len   */
    const int32_t *tokens;       /* device [n_rows] token ids                                                                 */
    const int32_t *row_slot;     /* device [n_rows] cache slot of every row                                                   */
    const int32_t *row_pos;      /* device [n_rows] position of every row's token                                             */
    const onebit_seg_t *segs;    /* HOST [n_seg]                                                                              */
    const int32_t *out_rows;     /* device [n_out]: the rows that sample (last row of an item whose prompt is complete)       */
    int32_t *next_tokens;        /* device [n_out] out                                                                        */
    void *logits;                /* optional fp16 [n_out, vocab]                                                              */
    float *part_val;             /* fp32 [ceil(vocab / 128) * 64] scratch                                                     */
    int32_t *part_idx;           /* int32 [ceil(vocab / 128) * 64] scratch                                                    */
    void *workspace;             /* onebit_mixed_workspace_bytes(...) bytes, 256-byte aligned, ZERO-FILLED ONCE               */
    size_t workspace_bytes;
} onebit_mixed_state_t;
/* max_dec_rows = state->n_slots of the steps that will use the workspace (the attention scratch is laid out for that many rows). */
size_t onebit_mixed_workspace_bytes(const onebit_model_t *model, int64_t max_rows, int32_t max_dec_rows, int32_t attn_chunk);
int onebit_mixed_step(const onebit_model_t *model, const onebit_mixed_state_t *state, void *stream);

/* One fused decode GEMV launch (the building block of onebit_decode_step, exposed so that a
 * single kernel can be measured and tested in isolation): up to 3 projections sharing the input
 * vector x, each writing its pre-LayerNorm u = fp16(fp16(W.(h*x)) * g) to outs[i] (fp16 [N_i]).
 * x is produced by the selected prologue:
 *   ONEBIT_PRO_PLAIN       x = xin
 *   ONEBIT_PRO_EMBED_RMS   r = embed[*token];                     x = RMSNorm(r) * rms_w
 *   ONEBIT_PRO_RES_LN_RMS  r = hres_in + LayerNorm(u_prev);       x = RMSNorm(r) * rms_w
 *   ONEBIT_PRO_SWIGLU      x = silu(LayerNorm(u_gate)) * LayerNorm(u_up)
 * (r is also written to hres_out when given).  All vectors fp16, length K = in_features. */
#define ONEBIT_PRO_PLAIN      0
#define ONEBIT_PRO_EMBED_RMS  1
#define ONEBIT_PRO_RES_LN_RMS 2
#define ONEBIT_PRO_SWIGLU     3
typedef struct onebit_fused_in {
    const void *xin, *embed, *hres_in, *u_prev, *rms_w, *u_gate, *u_up;
    const int32_t *token;
    void *hres_out;
    float rms_eps, ln_eps;
    /* optional per-tile LayerNorm partials (see onebit_decode_state_t.tile_stats), each
     * fp32 [ceil(n / 4096) * 512]: tile t of a vector holds (sum, M2) at [2t], [2t+1].
     * st_prev / st_gate / st_up: partials of u_prev / u_gate / u_up written by the launch that
     * produced them (all that the prologue reads must be given, or none: the statistics are then
     * recomputed from the vectors).  st_out[i]: where projection i publishes its own, or NULL.   */
    const float *st_prev, *st_gate, *st_up;
    float *st_out[3];
} onebit_fused_in_t;
int onebit_fused_gemv(const onebit_proj_t *projs, void *const *outs, int nproj, int prologue,
                      const onebit_fused_in_t *in, void *stream);

/* ---- train-mode layer: the reference's BitLinear + SignSTE on LATENT weights (bitnet.py:14-28, 58-68),
 * SURVEY.md section 8 rows a8 / f4 (forward and backward for knowledge-distillation training on MI355X).
 * All tensors of one call share `dtype` (ONEBIT_F16 / ONEBIT_F32); row-major, contiguous.
 *   forward   y[T,N] = LayerNorm( g * ( (x * h) . sign(W)^T ) ) (+ bias);   W [N,K] full precision, sign(0) = 0 (:18)
 *             z_save [T,N] (dtype) receives the GEMM output before * g, ln_stats [T,2] fp32 {mean, rstd}: what the
 *             backward pass reads instead of recomputing the GEMM.
 *   backward  from gy [T,N]: gx [T,K], gw [N,K] = (gz^T . (x*h)) * (1.001 - tanh(W)^2) (the STE, :21-23), gh [K],
 *             gg [N], gbias [N] (optional).  workspace: onebit_train_workspace_bytes(T,K,N,dtype), 16-byte aligned.
 * fp16: MFMA 16x16x16 f16 with fp32 accumulation, every tensor-level op of the reference rounded once to fp16;
 * fp32: MFMA 16x16x4 f32.  Deterministic (no atomics: column sums have a fixed order).                          */
size_t onebit_train_workspace_bytes(int64_t T, int64_t K, int64_t N, int dtype);
int onebit_train_forward(const void *x, const void *w_latent, const void *h, const void *g, const void *bias_or_null,
                         void *y, void *z_save, float *ln_stats, int64_t T, int64_t K, int64_t N, int dtype,
                         float ln_eps, void *stream);
int onebit_train_backward(const void *gy, const void *x, const void *w_latent, const void *h, const void *g,
                          const void *z_save, const float *ln_stats, void *gx, void *gw, void *gh, void *gg,
                          void *gbias_or_null, void *workspace, size_t workspace_bytes, int64_t T, int64_t K,
                          int64_t N, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ONEBIT_H */
