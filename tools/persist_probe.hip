// Skeleton of a PERSISTENT decode token (round 5): what would one launch per token cost on this chip if the five
// dependency edges of a decoder layer (q|k|v -> attention -> o -> gate|up -> down -> next layer) were hand-offs inside the
// launch instead of kernel boundaries?  The arithmetic is simulated (a dependent fma chain per phase, calibrated to the
// in-kernel times of tools/phase_probe.py); the DATA MOVEMENT is real and has the 7B shapes:
//   * every phase's producers publish their slice of the pre-LayerNorm vector + per-tile statistics, every consumer
//     workgroup gathers the whole vector (the reference's LayerNorm over a whole row makes every edge an all-to-all);
//   * the packed weights of a phase (q|k|v 32 KB, o 8, gate|up 48, down 22 KB per workgroup; 25 MB per layer) stream from
//     HBM into registers, requested ONE PHASE AHEAD of their use (the prefetch credit of MI355X_MICROARCH.md);
//   * 256 workgroups x 512 threads, one per CU, 32 layers per launch.
// Transports (MI355X_MICROARCH.md price list, Guideline 16 of cdna_hip_programming.md):
//   mode 0  8-byte {tag = epoch, value} granules, EVERY wave polls the granules of its own elements (registers)
//   mode 1  granules, ONE wave per workgroup sweeps the vector into LDS (16 loads in flight), barrier, everyone reads LDS
//   mode 3  write-through payload + one flag granule per producer; wave 0 polls the flags, then everyone loads the payload
//   mode 4  granules as mode 0, but wave 0 first polls ONE granule per producer (2 KB per round instead of the vector)
//   mode 9  the same skeleton as 160 dependent LAUNCHES in a hipGraph (plain stores / plain loads): the baseline
//   hybrid  [q|k|v -> attention -> o] as one launch with mode-0 hand-offs on its two SUBSET edges, gate|up and down as launches
// Every value is checked (hash of layer, phase, index); every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o persist_probe tools/persist_probe.hip && ./persist_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int NTH = 512, NPH = 5, MAXV = 22;      // MAXV: value granules per thread of the longest vector (11008 / 512)
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct PhaseDesc {
    int prod_first, prod_n;     // producing workgroups [first, first + n)
    int gpp, spp;               // value / statistics granules per producer
    int cons_first, cons_n;     // workgroups that consume this phase's output
    int subset;                 // > 0: a consumer reads only `subset` value granules (attention: its head's rows), all statistics
    int wloads;                 // 16-byte weight loads per lane that this phase's producers multiply
    int iters;                  // dependent fma chain simulating the phase's arithmetic
};
struct Params {
    PhaseDesc ph[NPH];
    u64 *gran[2];               // [layer parity]: value granules of the five phases at goff[], statistics at soff[]
    int goff[NPH], soff[NPH];
    u32 *plain;                 // modes 3 / 9: untagged payload (values at goff, statistics at soff), [parity]
    size_t plain_stride;
    const u32x4 *weights;       // [layers][256 workgroups][15 loads][512 lanes]
    int layers, use_weights, compute;
    u32 *err;                   // [0] wrong values, [1] timeouts
    u64 *stamps;                // [layers][NPH][3] wall clock (100 MHz) of workgroup stamp_wg: input ready, math done, published
    int stamp_wg;
    u32 spin_limit;
    u32 *sink;
};

__device__ __forceinline__ u32 val_of(int l, int p, int idx)
{
    return ((u32)(l + 1) * 0x9E3779B1u) ^ ((u32)(p + 1) * 0x85EBCA6Bu) ^ ((u32)idx * 0xC2B2AE35u);
}
__device__ __forceinline__ u64 now() { return wall_clock64(); }

struct WaveState { bool dead; u32 bad; };

// 16-byte device-scope loads (buffer_load_dwordx4 ... sc1: past the L1, like the 8-byte agent atomics) of granule PAIRS;
// one offset VGPR per thread, strides in the scalar offset, reads beyond the buffer return 0 (tag 0 = never an epoch)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00027000);
}
#define LD16(r, voff, soff) __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16)

__device__ __forceinline__ bool give_up(u32 spins, int lane, const Params &P, WaveState &ws)
{
    if (ws.dead || spins >= P.spin_limit) { if (!ws.dead && lane == 0) atomicAdd(P.err + 1, 1u); ws.dead = true; return true; }
    __builtin_amdgcn_s_sleep(1);
    return false;
}

// ---- mode 0: every wave polls the granules of its own elements -------------------------------------------------------------
// the layout of the decode kernels' prologue: thread t holds elements 8t .. 8t + 7 of every 4096-element vector = 4 granules
// = 32 contiguous bytes = two 16-byte loads; NV vectors (NV = 6: the SwiGLU prologue's two 11008-element inputs)
template <int NV>
__device__ __forceinline__ void poll_own(const u64 *g, int n, u32 epoch, int l, int p, int idx0, int tid, int lane, const Params &P, WaveState &ws)
{
    const __amdgpu_buffer_rsrc_t r = make_rsrc(g, n * 8);
    u32x4 v[NV][2];
    for (u32 spins = 0;; ++spins) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (k * 2048 < n) { v[k][0] = LD16(r, tid * 32, k * 16384); v[k][1] = LD16(r, tid * 32 + 16, k * 16384); }
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (k * 2048 < n) {
                const int i0 = k * 2048 + tid * 4;
                ok &= (i0 >= n || v[k][0][1] == epoch) && (i0 + 1 >= n || v[k][0][3] == epoch) && (i0 + 2 >= n || v[k][1][1] == epoch) && (i0 + 3 >= n || v[k][1][3] == epoch);
            }
        if (__all(ok)) break;
        if (give_up(spins, lane, P, ws)) return;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (k * 2048 < n) {
            const int i0 = k * 2048 + tid * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < n) ws.bad += v[k][j >> 1][2 * (j & 1)] != val_of(l, p, idx0 + i0 + j);
        }
}
// every WAVE reads all n <= 512 granules (tile statistics: each wave combines them redundantly, as ob_tiles_combine does):
// lane = 4 tiles = 8 granules = 64 contiguous bytes
__device__ __forceinline__ void poll_wave_all(const u64 *g, int n, u32 epoch, int l, int p, int idx0, int lane, const Params &P, WaveState &ws)
{
    const __amdgpu_buffer_rsrc_t r = make_rsrc(g, n * 8);
    u32x4 v[4];
    for (u32 spins = 0;; ++spins) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = LD16(r, lane * 64 + k * 16, 0);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) ok &= (lane * 8 + 2 * k >= n || v[k][1] == epoch) && (lane * 8 + 2 * k + 1 >= n || v[k][3] == epoch);
        if (__all(ok)) break;
        if (give_up(spins, lane, P, ws)) return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (lane * 8 + 2 * k < n) ws.bad += v[k][0] != val_of(l, p, idx0 + lane * 8 + 2 * k);
        if (lane * 8 + 2 * k + 1 < n) ws.bad += v[k][2] != val_of(l, p, idx0 + lane * 8 + 2 * k + 1);
    }
}

// ---- mode 1: one wave sweeps [g, g + n) into LDS, 16 loads of 16 bytes per lane and pass (2048 granules) ------------------------
__device__ __forceinline__ void sweep_to_lds(const u64 *g, int n, u32 epoch, u32 *dst, int lane, const Params &P, WaveState &ws)
{
    const __amdgpu_buffer_rsrc_t r = make_rsrc(g, n * 8);
    for (int base = 0; base < n; base += 2048) {
        u32x4 v[16];
        for (u32 spins = 0;; ++spins) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = LD16(r, lane * 16 + k * 1024, base * 8);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i0 = base + k * 128 + lane * 2;
                ok &= (i0 >= n || v[k][1] == epoch) && (i0 + 1 >= n || v[k][3] == epoch);
            }
            if (__all(ok)) break;
            if (give_up(spins, lane, P, ws)) return;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i0 = base + k * 128 + lane * 2;
            if (i0 < n) dst[i0] = v[k][0];
            if (i0 + 1 < n) dst[i0 + 1] = v[k][2];
        }
    }
}

// ---- the simulated arithmetic of a phase -----------------------------------------------------------------------------------
__device__ __forceinline__ float fake_math(float x, int ticks)          // busy for `ticks` of the 100 MHz wall clock
{
    const u64 t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < (long long)ticks) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    return x;
}

template <int N>
__device__ __forceinline__ void wload(u32x4 (&w)[N], const Params &P, int l, int b, int slot0, int tid)
{
    const u32x4 *base = P.weights + (((size_t)l * 256 + b) * 15 + slot0) * NTH + tid;
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __builtin_nontemporal_load(base + (size_t)i * NTH);
}
template <int N>
__device__ __forceinline__ u32 wuse(const u32x4 (&w)[N])
{
    u32 a = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) a ^= w[i][0] ^ w[i][1] ^ w[i][2] ^ w[i][3];
    return a;
}

// consume the output of phase pp (of layer lp) as workgroup b; returns when every value is on this CU
template <int MODE>
__device__ __forceinline__ void consume(const Params &P, int lp, int pp, int b, int tid, u32 *lds_vals, WaveState &ws)
{
    const PhaseDesc &D = P.ph[pp];
    if (b < D.cons_first || b >= D.cons_first + D.cons_n) return;
    const int lane = tid & 63, wave = tid >> 6;
    const u32 epoch = (u32)(lp * 8 + pp + 1);
    const int par = lp & 1;
    const int nval = D.prod_n * D.gpp, nst = D.prod_n * D.spp;
    const u64 *gv = P.gran[par] + P.goff[pp], *gs = P.gran[par] + P.soff[pp];
    // attention: the head's rows (a contiguous block of `subset` granules) and all statistics
    const int off = D.subset > 0 ? (b - D.cons_first) * D.subset : 0;
    const int nv = D.subset > 0 ? D.subset : nval;
    if (MODE == 4) {
        // wave 0 polls ONE granule per producer (its last), 2 KB per round instead of the whole vector; then every wave
        // sweeps its own granules (tags checked as in mode 0: the tail poll is a trigger, never the guarantee)
        if (wave == 0) {
            u64 v[4];
            for (u32 spins = 0;; ++spins) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __hip_atomic_load(gv + (size_t)min(k * 64 + lane, D.prod_n - 1) * D.gpp + (D.gpp - 1), RLX_AGENT);
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) ok &= (u32)(v[k] >> 32) == epoch;
                if (__all(ok)) break;
                if (give_up(spins, lane, P, ws)) break;
            }
        }
        __syncthreads();
    }
    if (MODE == 0 || MODE == 4) {
        poll_own<6>(gv + off, nv, epoch, lp, pp, off, tid, lane, P, ws);
        if (nst <= 512) poll_wave_all(gs, nst, epoch, lp, pp, 0x100000, lane, P, ws);      // redundantly per wave
        else poll_own<2>(gs, nst, epoch, lp, pp, 0x100000, tid, lane, P, ws);               // cooperative (SwiGLU form)
        __syncthreads();
    } else if (MODE == 1) {
        if (wave == 0) {
            sweep_to_lds(gv + off, nv, epoch, lds_vals, lane, P, ws);
            sweep_to_lds(gs, nst, epoch, lds_vals + 11264, lane, P, ws);
        }
        __syncthreads();
        if (!ws.dead || wave != 0) {
            for (int i = tid; i < nv; i += NTH) ws.bad += lds_vals[i] != val_of(lp, pp, off + i);
            for (int i = tid; i < nst; i += NTH) ws.bad += lds_vals[11264 + i] != val_of(lp, pp, 0x100000 + i);
        }
        __syncthreads();
    } else {   // MODE 3: flags
        if (wave == 0) {
            const u64 *gf = P.gran[par] + P.goff[pp];            // one flag granule per producer
            u64 v[4];
            for (u32 spins = 0;; ++spins) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __hip_atomic_load(gf + min(k * 64 + lane, D.prod_n - 1), RLX_AGENT);
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) ok &= (u32)(v[k] >> 32) == epoch;
                if (__all(ok)) break;
                if (give_up(spins, lane, P, ws)) break;
            }
        }
        __syncthreads();
        // payload: write-through stores on the producer side, device-scope (L1-bypassing) loads here: no acquire needed.
        // untagged: 4 values per 16 bytes; thread t holds values 8t .. 8t + 7 of every 4096-value block
        const u32 *pv = P.plain + par * P.plain_stride + P.goff[pp] * 2 + off, *ps = P.plain + par * P.plain_stride + P.soff[pp] * 2;
        const __amdgpu_buffer_rsrc_t rv = make_rsrc(pv, nv * 4), rs = make_rsrc(ps, nst * 4);
        u32x4 v[3][2], sv[2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k * 4096 < nv) { v[k][0] = LD16(rv, tid * 32, k * 16384); v[k][1] = LD16(rv, tid * 32 + 16, k * 16384); }
#pragma unroll
        for (int k = 0; k < 2; ++k) sv[k] = LD16(rs, tid * 32 + 16 * k, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k * 4096 < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k * 4096 + tid * 8 + j < nv) ws.bad += v[k][j >> 2][j & 3] != val_of(lp, pp, off + k * 4096 + tid * 8 + j);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (tid * 8 + j < nst) ws.bad += sv[j >> 2][j & 3] != val_of(lp, pp, 0x100000 + tid * 8 + j);
        __syncthreads();
    }
}

template <int MODE>
__device__ __forceinline__ void produce(const Params &P, int l, int p, int b, int tid)
{
    const PhaseDesc &D = P.ph[p];
    if (b < D.prod_first || b >= D.prod_first + D.prod_n) return;
    const u32 epoch = (u32)(l * 8 + p + 1);
    const int par = l & 1, j = b - D.prod_first;
    if (MODE == 3) {
        u32 *pv = P.plain + par * P.plain_stride + P.goff[p] * 2, *ps = P.plain + par * P.plain_stride + P.soff[p] * 2;
        if (tid < D.gpp) __hip_atomic_store(pv + j * D.gpp + tid, val_of(l, p, j * D.gpp + tid), RLX_AGENT);
        else if (tid < D.gpp + D.spp) __hip_atomic_store(ps + j * D.spp + (tid - D.gpp), val_of(l, p, 0x100000 + j * D.spp + (tid - D.gpp)), RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) __hip_atomic_store(P.gran[par] + P.goff[p] + j, ((u64)epoch << 32) | 1u, RLX_AGENT);
    } else {
        u64 *gv = P.gran[par] + P.goff[p], *gs = P.gran[par] + P.soff[p];
        if (tid < D.gpp) __hip_atomic_store(gv + j * D.gpp + tid, ((u64)epoch << 32) | val_of(l, p, j * D.gpp + tid), RLX_AGENT);
        else if (tid < D.gpp + D.spp)
            __hip_atomic_store(gs + j * D.spp + (tid - D.gpp), ((u64)epoch << 32) | val_of(l, p, 0x100000 + j * D.spp + (tid - D.gpp)), RLX_AGENT);
    }
}

// One launch per token.  Weight requests run one phase ahead of their use: o and gate|up rows are requested when q|k|v has been
// published (in flight under the attention), down rows after o, the next layer's q|k|v rows after gate|up.
template <int MODE>
__global__ __launch_bounds__(NTH) void persist_kernel(const Params P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32 *lds_vals = reinterpret_cast<u32 *>(smem);                       // [11264 + 2816]
    u64 *lds_stamp = reinterpret_cast<u64 *>(smem + 14336 * 4);          // [layers][5][3]
    const int b = blockIdx.x, tid = threadIdx.x;
    WaveState ws = {false, 0};
    float x = (float)tid;
    u32 acc = 0;
    const bool st = b == P.stamp_wg && tid == 0;
    u32x4 w0[4], w2[1], w3[6], w4[3];
    const bool in0 = b < P.ph[0].prod_n, in3 = b < P.ph[3].prod_n;
    if (P.use_weights && in0) wload(w0, P, 0, b, 0, tid);
#pragma unroll 1
    for (int l = 0; l < P.layers; ++l) {
#pragma unroll 1
        for (int p = 0; p < NPH; ++p) {
            const PhaseDesc &D = P.ph[p];
            if (p > 0 || l > 0) consume<MODE>(P, p == 0 ? l - 1 : l, p == 0 ? 4 : p - 1, b, tid, lds_vals, ws);
            if (st) lds_stamp[(l * 5 + p) * 3 + 0] = now();
            const bool prod = b >= D.prod_first && b < D.prod_first + D.prod_n;
            if (P.compute && prod) x = fake_math(x, D.iters);
            if (P.use_weights && prod) {
                if (p == 0) acc ^= wuse(w0);
                else if (p == 2) acc ^= wuse(w2);
                else if (p == 3) acc ^= wuse(w3);
                else if (p == 4) acc ^= wuse(w4);
            }
            if (st) lds_stamp[(l * 5 + p) * 3 + 1] = now();
            produce<MODE>(P, l, p, b, tid);
            if (st) lds_stamp[(l * 5 + p) * 3 + 2] = now();
            if (P.use_weights) {
                if (p == 0) { wload(w2, P, l, b, 4, tid); if (in3) wload(w3, P, l, b, 5, tid); }
                else if (p == 2) wload(w4, P, l, b, 11, tid);
                else if (p == 3 && in0 && l + 1 < P.layers) wload(w0, P, l + 1, b, 0, tid);
            }
        }
    }
    if (ws.bad) atomicAdd(P.err, ws.bad);
    if (st) for (int i = 0; i < P.layers * 15; ++i) P.stamps[i] = lds_stamp[i];
    if (x == 12345.678f || acc == 0x12345678u) P.sink[0] = acc + (u32)x;      // keep the math and the weight loads alive
}

// ---- mode 9: the same phase as its own launch (plain stores, plain loads after the kernel boundary) -----------------------------
__device__ __forceinline__ u32 consume_plain(const Params &P, int lp, int pp, int b, int tid)
{
    const PhaseDesc &S = P.ph[pp];
    if (b < S.cons_first || b >= S.cons_first + S.cons_n) return 0;
    const u32 *pv = P.plain + (lp & 1) * P.plain_stride + P.goff[pp] * 2, *ps = P.plain + (lp & 1) * P.plain_stride + P.soff[pp] * 2;
    int nval = S.prod_n * S.gpp, off = 0;
    const int nst = S.prod_n * S.spp;
    if (S.subset > 0) { off = (b - S.cons_first) * S.subset; nval = S.subset; }
    u32 v[MAXV], sv[8], bad = 0;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
        if (k * NTH < nval) v[k] = pv[off + min(k * NTH + tid, nval - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k * NTH < nst) sv[k] = ps[min(k * NTH + tid, nst - 1)];
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
        if (k * NTH < nval && k * NTH + tid < nval) bad += v[k] != val_of(lp, pp, off + k * NTH + tid);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k * NTH < nst && k * NTH + tid < nst) bad += sv[k] != val_of(lp, pp, 0x100000 + k * NTH + tid);
    return bad;
}
__device__ __forceinline__ void produce_plain(const Params &P, int l, int p, int b, int tid)
{
    const PhaseDesc &D = P.ph[p];
    if (b < D.prod_first || b >= D.prod_first + D.prod_n) return;
    const int j = b - D.prod_first;
    u32 *pv = P.plain + (l & 1) * P.plain_stride + P.goff[p] * 2, *ps = P.plain + (l & 1) * P.plain_stride + P.soff[p] * 2;
    if (tid < D.gpp) pv[j * D.gpp + tid] = val_of(l, p, j * D.gpp + tid);
    else if (tid < D.gpp + D.spp) ps[j * D.spp + (tid - D.gpp)] = val_of(l, p, 0x100000 + j * D.spp + (tid - D.gpp));
}

__global__ __launch_bounds__(NTH) void launch_kernel(const Params P, int l, int p)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const PhaseDesc &D = P.ph[p];
    float x = (float)tid;
    u32 acc = 0;
    const bool prod = b >= D.prod_first && b < D.prod_first + D.prod_n;
    const int pp = p == 0 ? 4 : p - 1, lp = p == 0 ? l - 1 : l;
    // input vectors first (the wait for them must not include the weight stream: the decode kernels' counted waits), then the
    // weight requests, which return underneath the arithmetic
    u32 bad = lp >= 0 ? consume_plain(P, lp, pp, b, tid) : 0;
    u32x4 w[6];
    const int slot0 = p == 0 ? 0 : (p == 2 ? 4 : (p == 3 ? 5 : 11));
    if (P.use_weights && prod && D.wloads > 0) {
        const u32x4 *base = P.weights + (((size_t)l * 256 + b) * 15 + slot0) * NTH + tid;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i < D.wloads) w[i] = __builtin_nontemporal_load(base + (size_t)i * NTH);
    }
    __syncthreads();
    if (P.compute && prod) x = fake_math(x, D.iters);
    if (P.use_weights && prod && D.wloads > 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i < D.wloads) acc ^= w[i][0] ^ w[i][1] ^ w[i][2] ^ w[i][3];
    }
    produce_plain(P, l, p, b, tid);
    if (bad) atomicAdd(P.err, bad);
    if (x == 12345.678f || acc == 0x12345678u) P.sink[0] = acc + (u32)x;
}

// ---- hybrid: q|k|v -> attention -> o as ONE launch (the two SUBSET edges in-launch: 192 producers -> 32 attention workgroups,
// 32 -> 256), the three big all-to-all edges (o -> gate|up -> down -> next layer) stay kernel boundaries.  o_proj's rows are
// requested at kernel entry (in flight under q|k|v and the attention).  Three launches per layer instead of five.
__global__ __launch_bounds__(NTH) void hybrid_kernel(const Params P, int l)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32 *lds_vals = reinterpret_cast<u32 *>(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    WaveState ws = {false, 0};
    float x = (float)tid;
    u32 acc = 0;
    const bool in0 = b < P.ph[0].prod_n;
    const bool attn = b >= P.ph[1].prod_first && b < P.ph[1].prod_first + P.ph[1].prod_n;
    u32 bad = l > 0 ? consume_plain(P, l - 1, 4, b, tid) : 0;
    u32x4 w0[4], w2[1];
    if (P.use_weights) { if (in0) wload(w0, P, l, b, 0, tid); wload(w2, P, l, b, 4, tid); }
    __syncthreads();
    if (P.compute && in0) x = fake_math(x, P.ph[0].iters);
    if (P.use_weights && in0) acc ^= wuse(w0);
    produce<0>(P, l, 0, b, tid);
    consume<0>(P, l, 0, b, tid, lds_vals, ws);
    if (P.compute && attn) x = fake_math(x, P.ph[1].iters);
    produce<0>(P, l, 1, b, tid);
    consume<0>(P, l, 1, b, tid, lds_vals, ws);
    if (P.compute) x = fake_math(x, P.ph[2].iters);
    if (P.use_weights) acc ^= wuse(w2);
    produce_plain(P, l, 2, b, tid);
    bad += ws.bad;
    if (bad) atomicAdd(P.err, bad);
    if (x == 12345.678f || acc == 0x12345678u) P.sink[0] = acc + (u32)x;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv)
{
    int layers = 32, reps = 7;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--layers") && i + 1 < argc) layers = atoi(argv[++i]);
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int G = 256;
    if (prop.multiProcessorCount < G) { printf("needs %d CUs\n", G); return 0; }
    Params P = {};
    // 7B shapes.  value granule = 2 fp16 rows, statistics: 2 fp32 per 16-row tile = 2 granules per tile
    //              producers        gpp spp  consumers       subset wloads iters
    P.ph[0] = {0, 192, 32, 8, 192, 32, 192, 4, 0};      // q|k|v: 192 workgroups x 4 tiles; consumed by 32 attention workgroups (idle in this phase)
    P.ph[1] = {192, 32, 64, 0, 0, 256, 0, 0, 0};        // attention: 32 heads x 128 outputs
    P.ph[2] = {0, 256, 8, 2, 0, 256, 0, 1, 0};          // o: 256 x 1 tile
    P.ph[3] = {0, 230, 48, 12, 0, 256, 0, 6, 0};        // gate|up: 230 x 6 tiles
    P.ph[4] = {0, 256, 8, 2, 0, 256, 0, 3, 0};          // down: 256 x 1 tile
    // simulated arithmetic per phase in us (prologue + MFMA + cross-wave sum with the weights already on the CU):
    // from tools/phase_probe.py stamps (round 4), weight wait removed
    const double math_us[NPH] = {2.0, 2.5, 1.0, 2.3, 3.5};
    int off = 0;
    for (int p = 0; p < NPH; ++p) { P.goff[p] = off; off += P.ph[p].prod_n * P.ph[p].gpp; P.soff[p] = off; off += P.ph[p].prod_n * P.ph[p].spp; off = (off + 63) & ~63; }
    const size_t ngran = off;
    for (int par = 0; par < 2; ++par) CK(hipMalloc(&P.gran[par], ngran * 8));
    P.plain_stride = ngran * 2;
    CK(hipMalloc(&P.plain, P.plain_stride * 2 * 4));
    const size_t wcount = (size_t)layers * 256 * 15 * NTH;
    u32x4 *wbuf;
    CK(hipMalloc(&wbuf, wcount * 16));
    CK(hipMemset(wbuf, 0x5a, wcount * 16));
    P.weights = wbuf;
    P.layers = layers;
    CK(hipMalloc(&P.err, 8)); CK(hipMalloc(&P.stamps, (size_t)layers * 15 * 8)); CK(hipMalloc(&P.sink, 4));
    P.stamp_wg = 5;
    P.spin_limit = 200000;
    printf("granules per parity %zu (%.1f KB), weights %.1f MB per layer\n", ngran, ngran * 8 / 1024.0, 256.0 * 15 * NTH * 16 / 1e6);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 14336 * 4 + (size_t)layers * 15 * 8 + 40 * 1024;      // > 80 KB: one workgroup per CU
    CK(hipFuncSetAttribute((const void *)persist_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)persist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)persist_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)persist_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    auto run_persist = [&](int mode, int use_w, int compute, bool print_tl) -> double {
        Params Q = P; Q.use_weights = use_w; Q.compute = compute;
        std::vector<double> t;
        u32 errs[2] = {0, 0};
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipMemsetAsync(Q.err, 0, 8, s));
            CK(hipEventRecord(e0, s));
            for (int par = 0; par < 2; ++par) CK(hipMemsetAsync(Q.gran[par], 0, ngran * 8, s));
            if (mode == 0) hipLaunchKernelGGL(persist_kernel<0>, dim3(G), dim3(NTH), lds, s, Q);
            else if (mode == 1) hipLaunchKernelGGL(persist_kernel<1>, dim3(G), dim3(NTH), lds, s, Q);
            else if (mode == 3) hipLaunchKernelGGL(persist_kernel<3>, dim3(G), dim3(NTH), lds, s, Q);
            else hipLaunchKernelGGL(persist_kernel<4>, dim3(G), dim3(NTH), lds, s, Q);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t.push_back(ms * 1e3);
            u32 e[2]; CK(hipMemcpy(e, Q.err, 8, hipMemcpyDeviceToHost));
            errs[0] += e[0]; errs[1] += e[1];
            if (e[1]) { printf("  mode %d: TIMEOUT (%u waves gave up) -- aborting this configuration\n", mode, e[1]); break; }
        }
        const double med = t.empty() ? -1.0 : median(t);
        printf("persistent mode %d weights %d math %d: %8.1f us per token-skeleton = %6.2f us per layer (min %.1f)  wrong values %u, timeouts %u\n",
               mode, use_w, compute, med, med / layers, t.empty() ? -1.0 : *std::min_element(t.begin(), t.end()), errs[0], errs[1]);
        if (print_tl && !t.empty()) {
            std::vector<u64> st((size_t)layers * 15);
            CK(hipMemcpy(st.data(), Q.stamps, st.size() * 8, hipMemcpyDeviceToHost));
            // average over layers 1.. of: edge (previous publish -> input ready), math, publish, per phase; 100 MHz clock
            double edge[NPH] = {}, math[NPH] = {}, pub[NPH] = {};
            for (int l = 1; l < layers; ++l)
                for (int p = 0; p < NPH; ++p) {
                    const u64 *c = &st[(l * 5 + p) * 3];
                    const u64 prev = p == 0 ? st[((l - 1) * 5 + 4) * 3 + 2] : st[(l * 5 + p - 1) * 3 + 2];
                    edge[p] += (double)(c[0] - prev) * 0.01; math[p] += (double)(c[1] - c[0]) * 0.01; pub[p] += (double)(c[2] - c[1]) * 0.01;
                }
            printf("  timeline of workgroup %d (us, mean over layers 1..%d): ", Q.stamp_wg, layers - 1);
            for (int p = 0; p < NPH; ++p) printf("[p%d wait %.2f math %.2f publish %.2f] ", p, edge[p] / (layers - 1), math[p] / (layers - 1), pub[p] / (layers - 1));
            printf("\n");
        }
        return med;
    };
    auto run_launches = [&](int use_w, int compute) -> double {
        Params Q = P; Q.use_weights = use_w; Q.compute = compute;
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int l = 0; l < layers; ++l)
            for (int p = 0; p < NPH; ++p) hipLaunchKernelGGL(launch_kernel, dim3(G), dim3(NTH), 0, s, Q, l, p);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        std::vector<double> t;
        u32 errs = 0;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipMemsetAsync(Q.err, 0, 8, s));
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t.push_back(ms * 1e3);
            u32 e[2]; CK(hipMemcpy(e, Q.err, 8, hipMemcpyDeviceToHost));
            errs += e[0];
        }
        const double med = median(t);
        printf("launches  (graph of %d) weights %d math %d: %8.1f us per token-skeleton = %6.2f us per layer (min %.1f)  wrong values %u\n",
               layers * NPH, use_w, compute, med, med / layers, *std::min_element(t.begin(), t.end()), errs);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
        return med;
    };

    auto run_hybrid = [&](int use_w, int compute) -> double {
        Params Q = P; Q.use_weights = use_w; Q.compute = compute;
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int par = 0; par < 2; ++par) CK(hipMemsetAsync(Q.gran[par], 0, ngran * 8, s));       // granule tags: once per token
        for (int l = 0; l < layers; ++l) {
            hipLaunchKernelGGL(hybrid_kernel, dim3(G), dim3(NTH), 60 * 1024, s, Q, l);
            hipLaunchKernelGGL(launch_kernel, dim3(G), dim3(NTH), 0, s, Q, l, 3);
            hipLaunchKernelGGL(launch_kernel, dim3(G), dim3(NTH), 0, s, Q, l, 4);
        }
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        std::vector<double> t;
        u32 errs[2] = {0, 0};
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipMemsetAsync(Q.err, 0, 8, s));
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t.push_back(ms * 1e3);
            u32 e[2]; CK(hipMemcpy(e, Q.err, 8, hipMemcpyDeviceToHost));
            errs[0] += e[0]; errs[1] += e[1];
            if (e[1]) break;
        }
        const double med = t.empty() ? -1.0 : median(t);
        printf("hybrid    (3 launches per layer: [q|k|v -> attention -> o] in-launch, gate|up, down) weights %d math %d: %8.1f us = %6.2f us per layer (min %.1f)  "
               "wrong values %u, timeouts %u\n", use_w, compute, med, med / layers, t.empty() ? -1.0 : *std::min_element(t.begin(), t.end()), errs[0], errs[1]);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
        return med;
    };
    double sum_math = 0;
    for (int p = 0; p < NPH; ++p) { P.ph[p].iters = (int)(math_us[p] * 100.0); sum_math += math_us[p]; }      // 100 MHz ticks
    printf("simulated arithmetic per layer: %.1f us (q|k|v %.1f, attention %.1f, o %.1f, gate|up %.1f, down %.1f), a wall-clock wait\n", sum_math, math_us[0],
           math_us[1], math_us[2], math_us[3], math_us[4]);

    printf("\n== floor: hand-offs only ==\n");
    run_launches(0, 0);
    run_hybrid(0, 0);
    for (int mode : {0, 1, 3, 4}) run_persist(mode, 0, 0, true);
    printf("\n== hand-offs + weight stream ==\n");
    run_launches(1, 0);
    run_hybrid(1, 0);
    for (int mode : {0, 1, 3, 4}) run_persist(mode, 1, 0, true);
    printf("\n== hand-offs + weight stream + simulated arithmetic ==\n");
    run_launches(1, 1);
    run_hybrid(1, 1);
    for (int mode : {0, 1, 3, 4}) run_persist(mode, 1, 1, true);
    printf("\n== arithmetic only (no weights) ==\n");
    run_launches(0, 1);
    run_hybrid(0, 1);
    for (int mode : {0, 1, 3, 4}) run_persist(mode, 0, 1, false);
    return 0;
}
