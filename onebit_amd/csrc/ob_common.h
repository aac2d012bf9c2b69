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
