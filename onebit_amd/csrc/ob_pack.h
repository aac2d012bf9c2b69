// Sign packer / unpacker kernels (HBM-bound byte work; one thread per packed byte).
//   pack  : scripts/convert_llama_to_infer_ckpt.py:7-15 applied to sign(w) (:30)
//   unpack: transformers/src/transformers/models/bitnet.py:98-110
#pragma once
#include "ob_common.h"

template <typename TW>
__global__ __launch_bounds__(256) void ob_pack_kernel(const TW *__restrict__ w,
                                                      uint8_t *__restrict__ packed, int64_t nbytes)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nbytes) return;
    TW v[8];
    if (sizeof(TW) == 2) {
        *reinterpret_cast<ob_u32x4 *>(v) = *reinterpret_cast<const ob_u32x4 *>(w + 8 * i);
    } else {
        reinterpret_cast<ob_u32x4 *>(v)[0] = reinterpret_cast<const ob_u32x4 *>(w + 8 * i)[0];
        reinterpret_cast<ob_u32x4 *>(v)[1] = reinterpret_cast<const ob_u32x4 *>(w + 8 * i)[1];
    }
    unsigned byte = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) byte |= (unsigned)((float)v[b] < 0.0f) << b;   // 0, -0, NaN -> bit 0 (+1)
    packed[i] = (uint8_t)byte;
}

// The reference's fp16_to_int8 itself (convert_llama_to_infer_ckpt.py:10-13) on ANY tensor, not only sign values:
// v = (0 - s + 1) / 2 in the tensor's dtype, truncated to uint8, byte = sum_i v_i * 2^i mod 256 (the uint8 matmul
// wraps) -- so |s| > 1 spills into the neighbouring bit positions exactly as there.  v < 1 (s > -1, incl. -0.5,
// NaN and the implementation-defined negative v of s > 1) contributes 0.
template <typename TW>
__global__ __launch_bounds__(256) void ob_f2i8_kernel(const TW *__restrict__ sgn, uint8_t *__restrict__ packed, int64_t nbytes)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nbytes) return;
    TW v[8];
    if (sizeof(TW) == 2) {
        *reinterpret_cast<ob_u32x4 *>(v) = *reinterpret_cast<const ob_u32x4 *>(sgn + 8 * i);
    } else {
        reinterpret_cast<ob_u32x4 *>(v)[0] = reinterpret_cast<const ob_u32x4 *>(sgn + 8 * i)[0];
        reinterpret_cast<ob_u32x4 *>(v)[1] = reinterpret_cast<const ob_u32x4 *>(sgn + 8 * i)[1];
    }
    unsigned byte = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const TW t = (TW)((TW)((TW)0 - v[b]) + (TW)1) / (TW)2;        // every op rounds in TW, as the torch expression does
        const float f = (float)t;
        const unsigned q = f >= 1.0f ? (unsigned)f : 0u;              // truncation toward zero; f <= 65504 / 2 fits
        byte += q << b;
    }
    packed[i] = (uint8_t)(byte & 0xffu);
}

template <typename TW>
__global__ __launch_bounds__(256) void ob_unpack_kernel(const uint8_t *__restrict__ packed,
                                                        TW *__restrict__ out, int64_t nbytes)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nbytes) return;
    const unsigned byte = packed[i];
    TW v[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = (TW)(1.0f - 2.0f * (float)((byte >> b) & 1u));
    if (sizeof(TW) == 2) {
        *reinterpret_cast<ob_u32x4 *>(out + 8 * i) = *reinterpret_cast<ob_u32x4 *>(v);
    } else {
        reinterpret_cast<ob_u32x4 *>(out + 8 * i)[0] = reinterpret_cast<ob_u32x4 *>(v)[0];
        reinterpret_cast<ob_u32x4 *>(out + 8 * i)[1] = reinterpret_cast<ob_u32x4 *>(v)[1];
    }
}
