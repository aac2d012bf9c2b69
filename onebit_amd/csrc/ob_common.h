// Shared device helpers for the OneBit gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef _Float16 ob_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 ob_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 ob_half2 __attribute__((ext_vector_type(2)));
typedef float ob_float4 __attribute__((ext_vector_type(4)));
typedef float ob_float2 __attribute__((ext_vector_type(2)));
typedef uint32_t ob_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ob_u32x2 __attribute__((ext_vector_type(2)));
typedef int32_t ob_i32x4 __attribute__((ext_vector_type(4)));

#define OB_WAVE 64

#if defined(OB_PROFILE_ABLATE) && !defined(OB_PROFILE_STAMPS)
#define OB_PROFILE_STAMPS 1      // the ablation build carries the phase timestamps too
#endif

// Wave64 reductions on the DPP network (row = 16 lanes): quad butterflies, row mirrors, then the two
// row broadcasts; ~8 VALU instructions, no LDS crossbar traffic (ds_bpermute-based shuffles cost
// ~100 cycles per step).  Every lane receives the result.
#define OB_DPP_F(v, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
__device__ __forceinline__ float ob_wave_sum(float v)
{
    v += OB_DPP_F(v, 0xB1, 0xF);     // quad_perm [1,0,3,2]
    v += OB_DPP_F(v, 0x4E, 0xF);     // quad_perm [2,3,0,1]
    v += OB_DPP_F(v, 0x141, 0xF);    // row_half_mirror
    v += OB_DPP_F(v, 0x140, 0xF);    // row_mirror: every lane of a row holds the row sum
    v += OB_DPP_F(v, 0x142, 0xA);    // row_bcast15 into rows 1, 3
    v += OB_DPP_F(v, 0x143, 0xC);    // row_bcast31 into rows 2, 3: row 3 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float ob_wave_max(float v)
{
    const int ninf = 0xff800000;
#define OB_DPP_M(v, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
    v = fmaxf(v, OB_DPP_M(v, 0xB1, 0xF));
    v = fmaxf(v, OB_DPP_M(v, 0x4E, 0xF));
    v = fmaxf(v, OB_DPP_M(v, 0x141, 0xF));
    v = fmaxf(v, OB_DPP_M(v, 0x140, 0xF));
    v = fmaxf(v, OB_DPP_M(v, 0x142, 0xA));
    v = fmaxf(v, OB_DPP_M(v, 0x143, 0xC));
#undef OB_DPP_M
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

typedef unsigned short ob_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t ob_wave_max_u32(uint32_t v)
{
#define OB_DPP_U(v, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false)
    v = max(v, OB_DPP_U(v, 0xB1, 0xF));
    v = max(v, OB_DPP_U(v, 0x4E, 0xF));
    v = max(v, OB_DPP_U(v, 0x141, 0xF));
    v = max(v, OB_DPP_U(v, 0x140, 0xF));
    v = max(v, OB_DPP_U(v, 0x142, 0xA));
    v = max(v, OB_DPP_U(v, 0x143, 0xC));
#undef OB_DPP_U
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Block-wide sum through LDS; `red` holds >= (blockDim.x / 64) floats.  All threads get the result.
__device__ __forceinline__ float ob_block_sum(float v, float *red)
{
    v = ob_wave_sum(v);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();                       // protect `red` from a previous use
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (nw == 4) return (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ float ob_round_h(float v) { return (float)(_Float16)v; }

// 16 sign bits (bit = 1 means -1) -> 8 dwords of packed fp16 (+-1.0) pairs.
// pair p holds weights (2p, 2p+1): even bits stay in the low half, odd bits are moved
// up by 15 so that one shift brings bit 2p to position 15 and bit 2p+1 to position 31.
__device__ __forceinline__ void ob_expand16(uint32_t bits16, uint32_t (&out)[8])
{
    const uint32_t c = (bits16 & 0x5555u) | ((bits16 & 0xAAAAu) << 15);
    // (x & mask) | (one & ~mask) with an opaque mask is the bitfield-insert pattern: ONE v_bfi_b32 per
    // dword instead of v_and + v_or with a literal each (the compiler folds the constant-mask form)
    uint32_t mask = 0x80008000u;
    asm("" : "+s"(mask));
#pragma unroll
    for (int p = 0; p < 8; ++p) out[p] = ((c << (15 - 2 * p)) & mask) | (0x3C003C00u & ~mask);
}

// ---------------------------------------------------------------------------------------------
// L2 prefetch of the NEXT launch's packed rows by the CUs a launch leaves idle.  The skinny GEMM launches of the batched
// step are short (5-14 us) and their first math waits for the first piece of packed rows from HBM (~2 us into the launch:
// transfers complete in order, so the L2-resident activation rows queue behind it); the row kernels before them run 32
// workgroups on a 256-CU chip.  Those launches carry one EXTRA workgroup per idle CU that does nothing but touch one dword
// of every 128-byte line of the consumer's rows -- the rows of the consuming workgroups on ITS OWN XCD (workgroups are dealt
// round-robin to the 8 XCDs, each with its own L2): the GEMM then finds its rows in L2 (measured with the rows resident:
// -1.0 us per launch).  A wrong guess of the placement costs nothing but the benefit.  The single-sequence decode step does
// the same for o_proj's rows from the attention launch (one workgroup per head).
// ---------------------------------------------------------------------------------------------
struct ObPfSeg {
    const char *base;             // first row of the segment (a projection, or a K-slice of one)
    long long ld_bytes;           // row pitch
    int row_bytes;                // bytes of a row that the consumer reads (K / 8 of the slice)
    int rows_per_wg, nrows;       // consumer workgroup b of [wg_begin, wg_end) reads rows (b - wg_begin) * rows_per_wg ...
    int wg_begin, wg_end;
};
struct ObPfPlan { ObPfSeg s[4]; int nseg; };

// x = XCD of this workgroup (its linear index in the grid & 7), j / nj = its rank among / the number of the grid's
// prefetching workgroups on that XCD; returns a value that depends on every loaded dword (so the loads are not dropped).
// WHO prefetches matters: a CU keeps ~64 missed lines (8 KB) in flight, i.e. 8-20 GB/s against HBM latency -- the 32
// working workgroups of a row kernel would need 10-20 us for a launch's 6-11 MB (measured: the step 2.27 -> 2.75 ms), and
// a kernel cannot end before its loads have returned (s_endpgm waits): dedicated workgroups on otherwise idle CUs.
__device__ __forceinline__ uint32_t ob_prefetch_l2(const ObPfPlan &P, int x, int j, int nj, int tid, int nthr)
{
    uint32_t acc = 0;
    for (int si = 0; si < P.nseg; ++si) {
        const ObPfSeg S = P.s[si];
        const int lines = (S.row_bytes + 127) >> 7;
        const int first = S.wg_begin + ((x - S.wg_begin) & 7);          // this XCD's consumers: first, first + 8, ...
        for (int b = first + 8 * j; b < S.wg_end; b += 8 * nj) {
            const int r0 = (b - S.wg_begin) * S.rows_per_wg;
            const int total = min(S.rows_per_wg, S.nrows - r0) * lines;
            for (int t = tid; t < total; t += nthr) {
                const int r = t / lines, l = t - r * lines;
                acc ^= *reinterpret_cast<const uint32_t *>(S.base + (long long)(r0 + r) * S.ld_bytes + min(l * 128, S.row_bytes - 4));
            }
        }
    }
    return acc;
}
// a grid of `nrows` working workgroups followed by prefetch-only ones: true (and done) for the latter
__device__ __forceinline__ bool ob_prefetch_only_wg(const ObPfPlan &P, int nrows, int tid, int nthr)
{
    const int w = (int)blockIdx.x;
    if (P.nseg <= 0 || w < nrows) return false;
    const int x = w & 7, first = nrows + ((x - nrows) & 7);
    const uint32_t acc = ob_prefetch_l2(P, x, (w - first) >> 3, ((int)gridDim.x - first + 7) >> 3, tid, nthr);
    asm volatile("" :: "v"(acc));
    return true;
}
