// Shared device helpers for the OneBit gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef _Float16 ob_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 ob_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 ob_half2 __attribute__((ext_vector_type(2)));
typedef float ob_float4 __attribute__((ext_vector_type(4)));
typedef uint32_t ob_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ob_u32x2 __attribute__((ext_vector_type(2)));
typedef int32_t ob_i32x4 __attribute__((ext_vector_type(4)));

#define OB_WAVE 64

__device__ __forceinline__ float ob_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float ob_wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
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
#pragma unroll
    for (int p = 0; p < 8; ++p) out[p] = ((c << (15 - 2 * p)) & 0x80008000u) | 0x3C003C00u;
}
