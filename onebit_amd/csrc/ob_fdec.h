// Decode attention over KEY BLOCKS (flash-decoding), gfx950: one query per (token row, head) against that row's KV-cache slot
// (modeling_bitllama.py:546-563: scores = q k^T / sqrt(D), softmax in fp32, probabilities . v), the positions split over
// blockIdx.z so that a long context is streamed by many workgroups instead of one (round 6; VERDICT r05 item 2: the batched
// step's one workgroup per (head, slot) read a 512-token context at ~15 GB/s and kept every score in LDS -- max_len <= ~15k).
//
// The query is GIVEN (LayerNorm + RoPE already applied, onebit_rows_qkv_rope_ragged, which also appended this token's key /
// value to the cache): grid = (head, row, split), 256 threads = 16 position groups x 16 lanes; a 16-lane DPP row spans the head
// dimension (8 halves per lane), so q.k is four v_dot2 + a row reduction and the probabilities never leave registers.  Every
// 16-lane row keeps its own running (max, sum, output[8]) over the positions it sweeps (online softmax, fp32); the rows are
// merged through LDS at the end of the split, the splits by the LAST workgroup of a (row, head) to arrive:
//   partials (m_s, l_s, o_s[D]) are published WRITE-THROUGH (relaxed agent-scope stores = sc1: no L2 write-back fence), every wave
//   drains vmcnt, one relaxed agent-scope ticket; the last arriver reads all partials with agent-scope loads and adds them in split
//   order -> deterministic whoever arrives last (cdna_hip_programming.md, split-K recipe, write-through form).  The ticket
//   counter returns to zero; the caller zero-fills the scratch once.
// Rounding: scores are rounded as the reference's eager attention does (fp16 matmul output, fp16 after / sqrt(D), :546); the
// probabilities are NOT rounded to fp16 before the value product (they are only known relative to the split's own maximum) --
// the same difference to the eager op order as the prefill flash kernel (ob_flash.h) and the reference's own
// LlamaFlashAttention2 switch (:588).  Bytes: 2 * 2 * D * L per (row, head): HBM-bound (8 loads of 16 B per thread in flight).
#pragma once
#include "ob_common.h"
#include "ob_rowstats.h"

struct ObFdecArgs {
    const _Float16 *q;            // [rows, H, D] post-RoPE queries (token-major)
    const _Float16 *k, *v;        // caches [slots][Hkv][max_len][D]; keys 0 .. row_pos[row] are valid
    _Float16 *o;                  // [rows, H * D]
    const _Float16 *h_next;       // optional [H * D]: o <- fp16(o * h_next) (o_proj's input scaling, bitnet.py:113)
    const int *row_slot;          // device [rows] or NULL (slot = row)
    const int *row_pos;           // device [rows]: position of the row's token; outside [0, max_len) = idle row
    float *part_o;                // [rows][H][nsplit][128] fp32 partial outputs (relative to the split's maximum)
    float *part_ml;               // [rows][H][nsplit][2]   {max, sum}
    int *counter;                 // [rows][H] arrival tickets, zero between launches
    int H, Hkv, D, max_len, n_slots, chunk, nsplit;
    float inv_sqrt_d;
    // FUSED form (the decode engines' key-block route without the separate rope / append launch): the query is formed HERE from the
    // pre-LayerNorm projection rows -- LayerNorm from the producer's per-tile partials (bitnet.py:118), optional bias (:119-120),
    // RoPE at the row's position (modeling_bitllama.py:175-181, every op rounded to fp16) -- by every (head, row, split) workgroup for
    // its own head (128 elements); the workgroup of the LAST live split also forms the new key and value, uses them for position
    // `pos` from LDS and -- one workgroup per kv head -- appends them to the cache (the arithmetic of ob_dec_attn_kernel's PST form).
    const _Float16 *u_q, *u_k, *u_v;     // [rows, H*D], [rows, Hkv*D] x 2
    const float *st_q, *st_k, *st_v;     // tile partials per row: ob_tile_stats_floats(n) floats each (ob_decode.h)
    const _Float16 *b_q, *b_k, *b_v;     // optional biases
    const _Float16 *cos, *sin;           // [max_pos, D]
    _Float16 *kw, *vw;                   // the caches again, writable
    float ln_eps;
};

__device__ __forceinline__ float ob_fd_row_sum(float v)     // sum over the 16 lanes of a DPP row, in every lane
{
    v += OB_DPP_F(v, 0xB1, 0xF);
    v += OB_DPP_F(v, 0x4E, 0xF);
    v += OB_DPP_F(v, 0x141, 0xF);
    v += OB_DPP_F(v, 0x140, 0xF);
    return v;
}
__device__ __forceinline__ float ob_fd_rows_sum(float v)    // over the wave's 4 rows (same lane & 15)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float ob_fd_rows_max(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

#define OB_FD_THREADS 256
#define OB_FD_PG 16               // position groups per workgroup

template <int NI, bool FUSED = false>   // NI: keys per thread in flight (16 * NI positions per sweep of the workgroup)
__global__ __launch_bounds__(OB_FD_THREADS) void ob_fdec_kernel(const ObFdecArgs A)
{
    __shared__ __attribute__((aligned(16))) float sm[4 * 128 + 16 + (FUSED ? 3 * 64 : 0)];      // po[4 waves][128] | red[16] | q, k, v (ONE LDS object)
    float *po = sm, *red = sm + 4 * 128;
    _Float16 *q_s = reinterpret_cast<_Float16 *>(sm + 4 * 128 + 16), *k_s = q_s + 128, *v_s = k_s + 128;
    const int head = blockIdx.x, row = blockIdx.y, split = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ds = tid & 15, pg = tid >> 4;
    const int slot = A.row_slot ? A.row_slot[row] : row;
    const int pos = A.row_pos[row];
    if (slot < 0 || slot >= A.n_slots || pos < 0 || pos >= A.max_len) return;        // idle row (uniform)
    const int L = pos + 1;
    const int p_lo = split * A.chunk;
    if (p_lo >= L) return;                                                           // empty split (uniform)
    const int p_hi = min(L, p_lo + A.chunk);
    const int nlive = (L + A.chunk - 1) / A.chunk;
    const int D = A.D, H = A.H;
    const int kvh = head / (H / A.Hkv);
    const bool dok = 8 * ds < D;
    const int dcl = dok ? 8 * ds : 0;
    const _Float16 *kb = A.k + (((int64_t)slot * A.Hkv + kvh) * A.max_len) * D + dcl;
    const _Float16 *vb = A.v + (((int64_t)slot * A.Hkv + kvh) * A.max_len) * D + dcl;
    // the next sweep's rows are requested before the current sweep's arithmetic (two register sets): a split of several sweeps
    // keeps 2 * NI loads per thread in flight instead of draining between sweeps
    ob_half8 k8[NI], v8[NI], kn[NI], vn[NI];
    auto fetch = [&](ob_half8 (&kd)[NI], ob_half8 (&vd)[NI], int base) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t off = (int64_t)min(base + pg + OB_FD_PG * i, L - 1) * D;
            kd[i] = __builtin_nontemporal_load(reinterpret_cast<const ob_half8 *>(kb + off));
            vd[i] = __builtin_nontemporal_load(reinterpret_cast<const ob_half8 *>(vb + off));
        }
    };
    fetch(k8, v8, p_lo);                                                             // in flight under the query's formation (FUSED)
    ob_half8 q8, kn8 = (ob_half8)(_Float16)0, vn8 = (ob_half8)(_Float16)0;
    const bool own_new = FUSED && split == nlive - 1;                                // (uniform) this split holds position `pos`
    if (FUSED) {
        // roles: wave 0 forms q, waves 1 / 2 (last live split only) k / v -- each combines ITS vector's tile partials; a lane holds
        // both elements of a rotate_half pair (d, d + D / 2)
        const int half = D >> 1, NQ = H * D, NK = A.Hkv * D;
        const int role = wave;
        if (role == 0 || (own_new && role < 3)) {                                    // (wave-uniform)
            const int n_r = role == 0 ? NQ : NK;
            const size_t sf = (size_t)((n_r + 4095) >> 12) * 512;
            const float *st_r = (role == 0 ? A.st_q : (role == 1 ? A.st_k : A.st_v)) + (size_t)row * sf;
            const _Float16 *ub = role == 0 ? A.u_q + (int64_t)row * NQ + head * D : (role == 1 ? A.u_k : A.u_v) + (int64_t)row * NK + kvh * D;
            const int d0 = min(lane, half - 1), d1 = d0 + half;
            ObTileStatsRt tr;
            ob_tiles_load_rt(tr, st_r, n_r, lane);
            const _Float16 ur0 = ub[d0], ur1 = ub[d1];
            _Float16 br0 = (_Float16)0, br1 = (_Float16)0;
            if (A.b_q) {                                                             // (uniform)
                const _Float16 *bb = role == 0 ? A.b_q + head * D : (role == 1 ? A.b_k : A.b_v) + kvh * D;
                br0 = bb[d0]; br1 = bb[d1];
            }
            const _Float16 rc0 = A.cos[(int64_t)pos * D + d0], rs0 = A.sin[(int64_t)pos * D + d0];
            const _Float16 rc1 = A.cos[(int64_t)pos * D + d1], rs1 = A.sin[(int64_t)pos * D + d1];
            float mr, rr;
            ob_tiles_combine_rt(tr, st_r, n_r, A.ln_eps, lane, mr, rr);
            float y0 = ob_ln_apply((float)ur0, mr, rr), y1 = ob_ln_apply((float)ur1, mr, rr);
            if (A.b_q) { y0 = ob_round_h(y0 + (float)br0); y1 = ob_round_h(y1 + (float)br1); }
            float e0 = y0, e1 = y1;
            if (role < 2) {                      // apply_rotary_pos_emb: x * cos + rotate_half(x) * sin, each op rounded to fp16
                e0 = ob_round_h(ob_round_h(y0 * (float)rc0) + ob_round_h(-y1 * (float)rs0));
                e1 = ob_round_h(ob_round_h(y1 * (float)rc1) + ob_round_h(y0 * (float)rs1));
            }
            _Float16 *dst = role == 0 ? q_s : (role == 1 ? k_s : v_s);
            if (lane < half) {
                dst[d0] = (_Float16)e0; dst[d1] = (_Float16)e1;
                if (role > 0 && head % (H / A.Hkv) == 0) {                            // one workgroup per kv head appends to the cache
                    _Float16 *cr = (role == 1 ? A.kw : A.vw) + (((int64_t)slot * A.Hkv + kvh) * A.max_len + pos) * D;
                    cr[d0] = (_Float16)e0; cr[d1] = (_Float16)e1;
                }
            }
            for (int d = D + lane; d < 128; d += 64) dst[d] = (_Float16)0;            // zero padding beyond the head dimension
        }
        __syncthreads();
        q8 = *reinterpret_cast<const ob_half8 *>(q_s + 8 * ds);
        if (own_new) { kn8 = *reinterpret_cast<const ob_half8 *>(k_s + 8 * ds); vn8 = *reinterpret_cast<const ob_half8 *>(v_s + 8 * ds); }
    } else {
        q8 = *reinterpret_cast<const ob_half8 *>(A.q + ((int64_t)row * H + head) * D + dcl);
        if (!dok) q8 = (ob_half8)(_Float16)0;
    }
    auto dot8 = [](const ob_half8 a, const ob_half8 b) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const ob_half2 x = {a[2 * e], a[2 * e + 1]}, y = {b[2 * e], b[2 * e + 1]};
            acc = __builtin_amdgcn_fdot2(x, y, acc, false);
        }
        return acc;
    };
    float m = -INFINITY, l = 0.f;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int base = p_lo; base < p_hi; base += OB_FD_PG * NI) {
        const bool more = base + OB_FD_PG * NI < p_hi;                               // (uniform)
        if (more) fetch(kn, vn, base + OB_FD_PG * NI);
        float sc[NI];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = base + pg + OB_FD_PG * i;
            // FUSED: the new token's key / value come from LDS, not from the cache (whose row `pos` another workgroup writes); the
            // masked tail of the last sweep (p > pos: clamped reads of that same row) takes them too -- finite values, weight 0
            if (FUSED && p >= pos) { k8[i] = kn8; v8[i] = vn8; }
            const float dot = ob_fd_row_sum(dot8(q8, k8[i]));
            const float sv = ob_round_h(ob_round_h(dot) * A.inv_sqrt_d);         // :546, fp16 matmul output, / sqrt(D) -> fp16
            sc[i] = p < p_hi ? sv : -INFINITY;
            mx = fmaxf(mx, sc[i]);
        }
        if (mx > m) {                                    // (uniform within a 16-lane row; -inf > -inf is false)
            const float f = __expf(m - mx);              // m = -inf: 0
            l *= f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= f;
            m = mx;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float pe = sc[i] == -INFINITY ? 0.f : __expf(sc[i] - m);
            l += pe;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(pe, (float)v8[i][e], o[e]);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NI; ++i) { k8[i] = kn[i]; v8[i] = vn[i]; }
        }
    }
    // ---- merge the workgroup's 16 rows: maximum, rescale, sums
    const float mw = ob_fd_rows_max(m);
    if (lane == 0) red[wave] = mw;
    __syncthreads();
    const float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));             // finite: position p_lo is valid
    const float f = m == -INFINITY ? 0.f : __expf(m - M);
    l = ob_fd_rows_sum(l * f);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ob_fd_rows_sum(o[e] * f);
    if (lane < 16) {
        float *dst = po + wave * 128 + 8 * ds;
        *reinterpret_cast<ob_float4 *>(dst) = (ob_float4){o[0], o[1], o[2], o[3]};
        *reinterpret_cast<ob_float4 *>(dst + 4) = (ob_float4){o[4], o[5], o[6], o[7]};
    }
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    float acc = 0.f;
    if (tid < 128) acc = (po[tid] + po[128 + tid]) + (po[256 + tid] + po[384 + tid]);
    const float lt = (red[4] + red[5]) + (red[6] + red[7]);
    _Float16 *orow = A.o + ((int64_t)row * H + head) * D;
    if (nlive == 1) {                                                                // (uniform) the whole context in one split
        if (tid < D) {
            _Float16 oh = (_Float16)(acc / lt);
            if (A.h_next) oh = oh * A.h_next[head * D + tid];
            orow[tid] = oh;
        }
        return;
    }
    // ---- publish this split's partial write-through, take a ticket; the last arriver combines in split order
    const int64_t rh = (int64_t)row * H + head;
    float *pso = A.part_o + (rh * A.nsplit) * 128;
    float *pml = A.part_ml + (rh * A.nsplit) * 2;
    if (tid < 128) __hip_atomic_store(pso + split * 128 + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 128) __hip_atomic_store(pml + split * 2, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 129) __hip_atomic_store(pml + split * 2 + 1, lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 // every wave: its stores have left
    __syncthreads();
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(A.counter + rh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == nlive - 1;
        if (last) __hip_atomic_store(A.counter + rh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[8] = last ? 1.f : 0.f;
    }
    __syncthreads();
    if (red[8] == 0.f) return;
    if (tid < D) {
        // every partial of the (row, head) requested at once (16 per batch: {max, sum} and this thread's output element), then a
        // fixed-order combine -- two dependent round trips (statistics, then outputs) would sit on the critical path of the step
        float Mg = -INFINITY, ls = 0.f, os = 0.f;
        for (int s0 = 0; s0 < nlive; s0 += 16) {
            float pm[16], pl[16], pv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int s = min(s0 + j, nlive - 1);
                pm[j] = __hip_atomic_load(pml + 2 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pl[j] = __hip_atomic_load(pml + 2 * s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pv[j] = __hip_atomic_load(pso + s * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            float mb = Mg;
#pragma unroll
            for (int j = 0; j < 16; ++j) mb = fmaxf(mb, pm[j]);                      // (clamped duplicates do not change the maximum)
            const float fo = Mg == -INFINITY ? 0.f : __expf(Mg - mb);                // rescale what earlier batches accumulated
            ls *= fo; os *= fo; Mg = mb;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (s0 + j < nlive) {
                    const float w = __expf(pm[j] - Mg);
                    ls = __builtin_fmaf(pl[j], w, ls);
                    os = __builtin_fmaf(pv[j], w, os);
                }
            }
        }
        _Float16 oh = (_Float16)(os / ls);
        if (A.h_next) oh = oh * A.h_next[head * D + tid];
        orow[tid] = oh;
    }
}
