// Causal prefill attention for gfx950 (modeling_bitllama.py:546-563: scores = q k^T / sqrt(D) + causal mask, softmax in
// fp32, probabilities . v), flash style: no [S, S] score tensor in HBM, fp32 online softmax, MFMA 16x16x32 f16.
// Replaces the vendor's fused attention (a Triton kernel behind torch SDPA) on the fused prefill route.
//
// Work decomposition: workgroup = (128 queries of one head of one sequence) x 4 waves, a wave owns 32 queries (two
// 16-query tiles) and sweeps the keys in blocks of 64 through LDS.  Everything is computed TRANSPOSED so that the
// probabilities never leave registers between the two matrix products:
//   S^T[key][query] = K . Q^T      A = K rows (lane: key = lane & 15, 8 consecutive d: one 16-byte LDS read of the
//                                  row-major K tile), B = Q^T (lane: query = lane & 15, 8 consecutive d: loaded once)
//                                  -> a lane holds, per 16-key tile, 4 consecutive keys of ONE query
//   O^T[d][query]  = V^T . P^T     B = P^T: the lane's 4 + 4 keys of two adjacent key tiles ARE its 8 k-elements (the
//                                  k order of a product is free as long as both operands agree), A = V^T rows
//                                  (lane: d = lane & 15, the same 8 keys).  Round 4: V stays ROW-MAJOR in LDS (stored like K:
//                                  16-byte pieces, conflict-free) and is transposed by the READ -- ds_read_b64_tr_b16: within a
//                                  16-lane group lane i supplies the address of chunk (key i >> 2, d 4 (i & 3) .. + 3) of a
//                                  4-key x 16-d block and lane c receives column c, the 4 keys of d = c (tools/tr16_probe.hip).
//                                  Two such reads are one A operand.  (Rounds 1-3 transposed in registers on the way in: 16
//                                  v_perm + eight 8-byte stores per thread and key block, 45 % of the LDS cycles in bank
//                                  conflicts -- profiles/r04_pmc_attention.txt.)
// so the softmax statistics of a query live in ONE lane column (lane & 15) across the 4 lane groups: row maxima / sums
// are in-lane reductions plus two row swaps (v_permlane16_swap / v_permlane32_swap), and the rescale factor of the
// running output is a per-lane scalar.  Output: 4 consecutive d of one query per lane and tile, written token-major
// [B, S, H, D] = the rows o_proj consumes, optionally already multiplied by o_proj's input_factor (bitnet.py:113) so
// that the projection runs with ONEBIT_FLAG_PRESCALED.
// Causality: key blocks above the diagonal are never loaded; diagonal blocks are masked per element (a separate instance of the
// block body, so the blocks below the diagonal carry no mask code); a workgroup takes query block n - 1 - j and then block j of
// its (sequence, head), so every workgroup sweeps the same number of key blocks.
// Round 4 (546 -> 654 TFLOP/s at 8 x 2048 x 32 x 128; every step measured with tools/flash_lab.hip):
//   * K and V tiles double-buffered in LDS, ONE barrier per key block; rows unpadded and XOR-swizzled -- no bank conflict left
//     (SQ_LDS_BANK_CONFLICT 0, was 45 % of the LDS cycles)
//   * K / V pieces by buffer loads: lane-constant offset + one add per block, rows past the last key read as zeros
//   * the running maximum moves only when a tile exceeds it by more than 2^8 (the rescale of the output was taken for half the
//     blocks on random scores)
//   * issue order pinned where hipcc's own choice left LDS latency in front of the matrix pipe (K fragments 2 tiles ahead, V
//     fragments of the first 32 keys requested before the softmax arithmetic)
// What bounds it now (ablations in flash_lab: no softmax arithmetic 755, no staging 821, neither 1039 TFLOP/s): the softmax VALU
// work and the K / V staging of a wave run in series with its MFMAs, and the second wave of the SIMD covers little of it.  Three
// rearrangements meant to force the overlap were built, are correct, and measured SLOWER; their numbers are in
// docs/experiments.md, their code in the git history (tools/attic, removed in round 5): one wave per SIMD with 64 queries per wave and hand-interleaved softmax (565; ob_flash64_experiment.h), an
// 8-wave ping-pong where one wave of a SIMD streams MFMAs while its partner does only VALU work (599; ob_flash_pp.h -- its
// ablations show MFMA-phase time + softmax-phase time ~ total time, although pure MFMA and VALU streams of two waves do overlap
// on this SIMD: tools/pipe_overlap_probe.hip, profiles/r04_pipe_overlap_probe.txt), and
// this kernel with the two query tiles of a wave taken in turn, one tile's exponentials interleaved with the other's MFMAs
// (577; ob_flash_split_experiment.h).
#pragma once
#include <type_traits>
#include "ob_common.h"

struct ObFlashArgs {
    const _Float16 *q;        // [B, S, H, D] token-major (onebit_rows_qkv_rope with ONEBIT_FLAG_Q_TOKEN_MAJOR)
    const _Float16 *k, *v;    // cache rows [B][Hkv][max_len][D]; keys 0 .. past + S - 1 are valid
    _Float16 *o;              // [B, S, H, D]
    const _Float16 *h_next;   // optional [H * D]: o <- fp16(o * h_next)
    int S, H, Hkv, max_len, past;
    float scale_log2e;        // log2(e) / sqrt(D)
    int nmb;                  // query blocks per (batch, head)
#ifdef OB_FL_TRACE
    unsigned long long *trace;    // tools/flash_lab.hip: s_memtime stamps of two workgroups
#endif
};

// RAGGED form (round 6, the mixed prefill + decode step of continuous batching): the queries are the token rows of SEVERAL
// sequences, concatenated; segment i = rows [row0, row0 + n) of the request in KV-cache slot `slot`, first new token at position
// `past` (its keys 0 .. past + n - 1 are in the cache).  The segment table travels in the kernel arguments (<= 64 segments per
// launch: no device-side table to upload, graph-capture safe); wg_end[i] = workgroups of segments 0 .. i (n_heads * ceil(query
// blocks / 2) each).  Everything else is the kernel above with (b, S, past) read per segment.
#define OB_FL_MAXSEG 64
struct ObFlashSeg { int row0, n, slot, past; };
struct ObFlashRaggedArgs {
    ObFlashArgs a;            // q / o [rows, H, D]; k / v [slots][Hkv][max_len][D]; S, past, nmb unused
    int nseg;
    int wg_end[OB_FL_MAXSEG];
    ObFlashSeg seg[OB_FL_MAXSEG];
};

#define OB_FL_BM 128
#define OB_FL_BN 64
#define OB_FL_THREADS 256
#ifdef OB_FL_TRACE
#define OB_FL_T(i) do { if (tr) tr[(pass * 32 + kb) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define OB_FL_TP(i) do { if (tr) tr[(64 + pass) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OB_FL_T(i) do { } while (0)
#define OB_FL_TP(i) do { } while (0)
#endif

__device__ __forceinline__ float ob_fl_col_max(float v)      // max over the 4 lanes that share lane & 15
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float ob_fl_col_sum(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

#ifndef OB_FL_ABL
#define OB_FL_ABL 0             // tools/flash_lab.hip ablations (timing only): 1 = no softmax arithmetic, 2 = no K / V staging in the loop
#endif
#ifndef OB_FL_DEFER_THR
#define OB_FL_DEFER_THR 8.0f    // log2 units: the running maximum is kept while the new tile's maximum exceeds it by less (P <= 2^8)
#endif

template <int D, bool RAGGED = false>
__global__ __launch_bounds__(256, 2) void ob_flash_fwd_kernel(const std::conditional_t<RAGGED, ObFlashRaggedArgs, ObFlashArgs> AA)
{
    const ObFlashArgs &A = [&]() -> const ObFlashArgs & { if constexpr (RAGGED) return AA.a; else return AA; }();
    constexpr int DT = D / 16, DK = D / 32;
    constexpr int NPC = D / 8;                  // 16-byte pieces per K / V row
    constexpr int KLD = OB_FL_BN * NPC / 256;   // K (and V) pieces per thread and block (4 at D = 128)
    constexpr int RPL = 256 / NPC;              // key rows one pass of the 256 threads covers
    // K and V tiles: two buffers each, rows UNPADDED (one buffer_load_dwordx4 per 16-byte piece, 16 lanes a row) and the pieces
    // of a row XOR-swizzled by the row so that the MFMA operand reads are conflict-free:
    //   K piece pc of row r sits at pc ^ kswz(r): a ds_read_b128 lane group (rows {0-3, 12-15} of lane group g, rows 4-11 of
    //     g + 1; MI355X_MICROARCH LDS table) then touches 16 different pieces -- a 16-byte row pad cannot do that (rows r, g and
    //     r + 1, g - 1 share banks: the 2-way conflicts of rounds 1-3)
    //   V piece pc of row r sits at pc ^ 2 (r mod NPC / 2): the 8 rows a half-wave of the transpose read touches (32 B each)
    //     fall on disjoint banks
    __shared__ __attribute__((aligned(16))) _Float16 Ks[2][OB_FL_BN][D];
    __shared__ __attribute__((aligned(16))) _Float16 Vs[2][OB_FL_BN][D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    // XCD-aware numbering: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so
    // the logical block ids are renumbered to give every XCD one CONTIGUOUS range -- the 16 query blocks of a (sequence,
    // head) then run on one XCD, close in time, and its K / V rows (1 MB at S = 2048) are fetched from HBM once instead
    // of once per XCD and query block (measured: 1.02 -> see DESIGN.md; dealt plainly, 8 x 2048 x 32 heads re-read
    // 2.2 GB of K / V through thrashing L2s).  Within a head: heaviest query blocks first (they sweep the most keys).
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, rem = nwg & 7;
    const int bid = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (orig >> 3);
    // Causal balance: a workgroup takes query block npair - 1 - j .. and then block j of its (sequence, head) -- every
    // workgroup sweeps the same number of key blocks (heaviest-first dealing of single blocks left a 12 % tail)
    int npair, pj, head, b, S, past, nmb;
    int64_t qrow0;                                // first token row of this sequence's queries in q / o
    if constexpr (RAGGED) {
        // segment of this workgroup: lane l compares with wg_end[l] (one vector load of the kernarg table, one ballot)
        const int we = lane < AA.nseg ? AA.wg_end[lane] : 0x7fffffff;
        const int sg = __builtin_popcountll(__builtin_amdgcn_ballot_w64(bid >= we));
        const int wg0 = sg ? AA.wg_end[sg - 1] : 0;
        const ObFlashSeg sgd = AA.seg[sg];
        S = sgd.n; past = sgd.past; b = sgd.slot; qrow0 = sgd.row0;
        nmb = (S + OB_FL_BM - 1) / OB_FL_BM;
        npair = (nmb + 1) >> 1;
        const int loc = bid - wg0;
        pj = loc % npair; head = loc / npair;
    } else {
        S = A.S; past = A.past; nmb = A.nmb;
        npair = (nmb + 1) >> 1;
        pj = bid % npair;
        const int bh = bid / npair;
        head = bh % A.H; b = bh / A.H;
        qrow0 = (int64_t)b * S;
    }
    const int kvh = head / (A.H / A.Hkv);
    const int L = past + S;
    // K / V rows of this head as buffer resources: the offset of a piece is (lane-constant) + (block-uniform scalar), no per-block
    // address arithmetic, and rows beyond the last key read as zeros (they are masked; zeros keep 0 x V finite)
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.k + ((int64_t)b * A.Hkv + kvh) * A.max_len * D), 0, L * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.v + ((int64_t)b * A.Hkv + kvh) * A.max_len * D), 0, L * D * 2, 0x00020000);

#ifdef OB_FL_TRACE
    unsigned long long *tr = nullptr;
    if (A.trace && (orig == 0 || orig == nwg / 2) && lane == 0) tr = A.trace + ((orig ? 1 : 0) * (OB_FL_THREADS / 64) + wave) * 66 * 8;
#endif
    for (int pass = 0; pass < 2; ++pass) {
    OB_FL_TP(0);
    const int mb = pass == 0 ? nmb - 1 - pj : pj;
    if (pass == 1 && mb == nmb - 1 - pj) break;                 // odd count: the middle block has no partner
    const int m0 = mb * OB_FL_BM;
    // Q^T fragments of the wave's two query tiles: lane (query = lr, d = 32 ds + 8 g .. + 7)
    ob_half8 qf[2][DK];
    int qpos[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int s = m0 + 32 * wave + 16 * qt + lr;
        qpos[qt] = past + s;                                    // absolute position of this lane's query
        const _Float16 *qr = A.q + ((qrow0 + min(s, S - 1)) * A.H + head) * D;
#pragma unroll
        for (int ds = 0; ds < DK; ++ds) qf[qt][ds] = *reinterpret_cast<const ob_half8 *>(qr + 32 * ds + 8 * g);
    }
    ob_float4 acc_o[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc_o[dt][qt] = (ob_float4){0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    // key blocks this workgroup needs: keys 0 .. past + min(m0 + 128, S) - 1
    const int last_q = past + min(m0 + OB_FL_BM, S) - 1;
    const int nkb = last_q / OB_FL_BN + 1;
    const int wave_last_q = past + min(m0 + 32 * wave + 31, S - 1);        // beyond it every key is masked for this wave

    // staging through registers: thread t moves piece t % NPC of rows t / NPC + RPL i
    ob_u32x4 kreg[KLD], vreg[KLD];
    const int srow = tid / NPC, spc = tid % NPC;
    const int kswz_s = NPC == 16 ? (srow & 15) : ((srow >> 1) & 7);        // RPL is a multiple of 16: the same for every i
    const int vswz_s = (srow & (NPC / 2 - 1)) << 1;
    auto store_piece = [&](int buf, int j) {    // j < KLD: K piece j; else V piece j - KLD
        if (j < KLD) *reinterpret_cast<ob_u32x4 *>(&Ks[buf][srow + RPL * j][8 * (spc ^ kswz_s)]) = kreg[j];
        else *reinterpret_cast<ob_u32x4 *>(&Vs[buf][srow + RPL * (j - KLD)][8 * (spc ^ vswz_s)]) = vreg[j - KLD];
    };
    auto load_piece = [&](int kb, int j) {
        const int vo = tid * 16 + kb * OB_FL_BN * D * 2;
        if (j < KLD) kreg[j] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, vo + j * RPL * D * 2, 0, 0));
        else vreg[j - KLD] = __builtin_bit_cast(ob_u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, vo + (j - KLD) * RPL * D * 2, 0, 0));
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2 * KLD; ++j) store_piece(buf, j);
    };
    auto load_block = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 2 * KLD; ++j) load_piece(kb, j);
    };
    // operand addresses inside a tile (halves): K row lr of a 16-key tile, piece 4 ds + g; V (transpose read: this lane's chunk of
    // a 4-key x 16-d block = row lr >> 2 of the block, columns 4 (lr & 3) .. + 3; lane group g's block starts at key 4 g)
    typedef short ob_v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) ob_v4s ob_lds_v4s;
    const int kswz_r = NPC == 16 ? lr : ((lr >> 1) & 7);
    int koff[DK], voff[DT];
#pragma unroll
    for (int ds = 0; ds < DK; ++ds) koff[ds] = lr * D + 8 * ((4 * ds + g) ^ kswz_r);
    const int vrow = 4 * g + (lr >> 2), vswz_r = vrow & (NPC / 2 - 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) voff[dt] = vrow * D + 8 * (2 * (dt ^ vswz_r) + ((lr & 3) >> 1)) + 4 * (lr & 1);

    // Pipeline: block kb is computed from LDS buffer kb & 1; block kb + 1 (in registers since the previous iteration) goes to the
    // other buffer first thing (its last readers passed the barrier that ended iteration kb - 1), which frees the staging
    // registers for the K fragments; block kb + 2 leaves memory once the score MFMAs are issued.  One barrier per block.
    // Issue order inside a block is pinned where the compiler's own choice serialised LDS latency with the matrix pipe
    // (it kept 2-4 fragment reads in flight): ALL K fragments are requested before the first score MFMA, the V fragments of the
    // first 32 keys before the softmax arithmetic, those of the other 32 before the first output MFMA.
    typedef short ob_v8s __attribute__((ext_vector_type(8)));
    auto block = [&](const int kb, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;     // the blocks that reach a diagonal or the end of the keys (masks, idle waves)
        OB_FL_T(0);
        const int k0 = kb * OB_FL_BN, buf = kb & 1;
        const bool active = !TAIL || k0 <= wave_last_q;     // (wave-uniform) otherwise every key of this block is masked for this wave
        const bool diag = TAIL && (k0 + OB_FL_BN - 1 > past + m0 + 32 * wave || k0 + OB_FL_BN > L);
        const _Float16 *Kb = &Ks[buf][0][0], *Vb = &Vs[buf][0][0];
        const bool do_store = !(OB_FL_ABL & 2) && (!TAIL || kb + 1 < nkb);   // (past the last block the loads returned zeros: harmless)
        const bool do_load = !(OB_FL_ABL & 2) && (!TAIL || kb + 2 < nkb);
        ob_float4 sc[4][2];
        if (active) {
            // ---- S^T = K . Q^T: 4 key tiles x 2 query tiles, K = D.  One staging store rides behind every fourth MFMA.
            // K fragments: 2 DK reads ahead of the MFMAs that consume them (all 4 DK at once would not fit beside the staging registers)
            ob_half8 kf[4][DK];
            auto read_k = [&](int i) { kf[i / DK][i % DK] = *reinterpret_cast<const ob_half8 *>(Kb + 16 * (i / DK) * D + koff[i % DK]); };
#pragma unroll
            for (int i = 0; i < 2 * DK; ++i) read_k(i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) sc[kt][qt] = (ob_float4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4 * DK; ++i) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) sc[i / DK][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i / DK][i % DK], qf[qt][i % DK], sc[i / DK][qt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 2 * DK < 4 * DK) read_k(i + 2 * DK);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (do_store) store_block(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        } else if (do_store) store_block(buf ^ 1);
        OB_FL_T(1);
        if (do_load) load_block(kb + 2);
        OB_FL_T(2);
        if (active) {
            ob_half8 vf[2][DT];
            auto read_v = [&](int ks) {     // A = the 4 keys of key tile 2 ks, then of tile 2 ks + 1, for d = 16 dt + lr (two transpose reads)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const ob_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(Vb + 32 * ks * D + voff[dt]));
                    const ob_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(Vb + (32 * ks + 16) * D + voff[dt]));
                    const ob_v8s a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    vf[ks][dt] = __builtin_bit_cast(ob_half8, a8);
                }
            };
            read_v(0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- causal mask (only blocks that reach this wave's diagonal or the end of the keys), online softmax
            ob_half8 pb[2][2];
#if OB_FL_ABL & 1
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pb[qt][ks][e] = (_Float16)sc[2 * ks][qt][e];
                        pb[qt][ks][4 + e] = (_Float16)sc[2 * ks + 1][qt][e];
                    }
            l_run[0] = l_run[1] = 1.f;
#else
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                if (TAIL && diag) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kidx = k0 + 16 * kt + 4 * g + e;
                            if (kidx > qpos[qt] || kidx >= L) sc[kt][qt][e] = -INFINITY;
                        }
                }
                float mx = sc[0][qt][0];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sc[kt][qt][e]);
                mx = ob_fl_col_max(mx);
                // The running maximum moves (and the running output and sum are rescaled) only when some query of the tile
                // sees a maximum more than 2^THR above it -- a wave-uniform branch that random and real scores alike take
                // for the first block or two only.  Exact arithmetic otherwise: P = exp(s - m_run) <= 2^THR is still a
                // normal fp16 number with the same relative rounding, and the sums are fp32.  (-inf - -inf = NaN compares
                // false: a row with every key masked so far stays at m_run = -inf and P = 0.)
                if (__builtin_amdgcn_ballot_w64((mx - m_run[qt]) * A.scale_log2e > OB_FL_DEFER_THR) != 0) {
                    const float m_new = fmaxf(m_run[qt], mx);
                    const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m_run[qt] - m_new) * A.scale_log2e);
                    l_run[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) acc_o[dt][qt] *= alpha;
                    m_run[qt] = m_new;
                }
                const float nm = m_run[qt] == -INFINITY ? 0.f : -m_run[qt] * A.scale_log2e;
                float ls = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][qt][e], A.scale_log2e, nm));
                        sc[kt][qt][e] = p;
                        ls += p;
                    }
                l_run[qt] += ls;
                // P^T operands: k-elements of step ks = this lane's 4 keys of tile 2 ks, then of tile 2 ks + 1
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pb[qt][ks][e] = (_Float16)sc[2 * ks][qt][e];
                        pb[qt][ks][4 + e] = (_Float16)sc[2 * ks + 1][qt][e];
                    }
            }
#endif
            OB_FL_T(3);
            // ---- O^T += V^T . P^T
            read_v(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) acc_o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[ks][dt], pb[qt][ks], acc_o[dt][qt], 0, 0, 0);
        }
        OB_FL_T(4);
        __syncthreads();                        // block kb + 1 is complete in the other buffer; this one may be overwritten
    };
    const int nfull = min((past + m0 + 1) / OB_FL_BN, nkb);     // blocks entirely below every wave's diagonal
    load_block(0);
    store_block(0);
    load_block(1);
    __syncthreads();
    OB_FL_TP(1);
    int kb = 0;
    for (; kb < nfull; ++kb) block(kb, std::false_type());
    for (; kb < nkb; ++kb) block(kb, std::true_type());

    OB_FL_TP(2);
    // ---- normalise and write: lane holds d = 16 dt + 4 g .. + 3 of query lr of each tile
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int s = m0 + 32 * wave + 16 * qt + lr;
        const float l = ob_fl_col_sum(l_run[qt]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (s >= S) continue;
        _Float16 *orow = A.o + ((qrow0 + s) * A.H + head) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            ob_half4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (_Float16)(acc_o[dt][qt][e] * inv);
            if (A.h_next) ov = ov * *reinterpret_cast<const ob_half4 *>(A.h_next + head * D + 16 * dt + 4 * g);
            *reinterpret_cast<ob_half4 *>(orow + 16 * dt + 4 * g) = ov;
        }
    }
    OB_FL_TP(3);
    }
}
