// Per-CU rate of pulling an L2-resident [32, 4096] fp16 block (256 KB, the batched step's activation rows) into a CU,
// every workgroup reading ALL of it (what the skinny 1-bit GEMMs do): LDS-DMA (global_load_lds_dwordx4) against
// ordinary 16-byte loads (+ ds_write_b128), 4 / 8 / 16 waves per CU, rows of 8 x 128 B per instruction (the GEMM's
// access) or 1 KB contiguous.   hipcc --offload-arch=gfx950 -O3 -o tools/l2_rate_probe tools/l2_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NW, int CONTIG>
__global__ __launch_bounds__(64 * NW) void probe(const char *a, int passes, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // instruction j of the block (256 of them, 1 KB each): CONTIG: bytes [1024 j, +1024); else piece p = j / 8 (256 B of
    // every row), d = j % 8: rows 8 (d % 4) .. + 7 of sub-tile d / 4 (128 B)
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + wave * 8192;
    u32x4 acc = {0, 0, 0, 0};
    const int rot = (blockIdx.x >> 3) & 31;
    for (int ps = 0; ps < passes; ++ps) {
        for (int j0 = wave * 8; j0 < 256; j0 += NW * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int j = j0 + u;
                uint32_t off;
                if (CONTIG) off = (uint32_t)(((j + 8 * rot) & 255) * 1024 + lane * 16);
                else {
                    const int p = ((j >> 3) + rot) & 31, d = j & 7;
                    off = (uint32_t)((8 * (d & 3) + (lane >> 3)) * 8192 + p * 256 + (d >> 2) * 128 + (lane & 7) * 16);
                }
                if (MODE == 0) {
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(lds0 + u * 1024), "v"(off), "s"(a) : "memory");
                } else {
                    v[u] = *reinterpret_cast<const u32x4 *>(a + off);
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u];
            }
            if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < 8; ++u) *reinterpret_cast<u32x4 *>(smem + wave * 8192 + u * 1024 + lane * 16) = v[u];
            }
        }
    }
    if (MODE == 2 || MODE == 0) acc = *reinterpret_cast<u32x4 *>(smem + wave * 8192 + lane * 16);
    if (acc[0] == 0x12345678u) sink[threadIdx.x] = acc[1] ^ acc[2] ^ acc[3];
}

template <int MODE, int NW, int CONTIG>
static void run(const char *a, uint32_t *sink, const char *name)
{
    const int passes = 40;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)probe<MODE, NW, CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 8192);
    hipLaunchKernelGGL((probe<MODE, NW, CONTIG>), dim3(256), dim3(64 * NW), NW * 8192, 0, a, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NW, CONTIG>), dim3(256), dim3(64 * NW), NW * 8192, 0, a, passes, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * 262144.0 * passes;
    printf("%-46s %2d waves/CU  %s   %7.1f us per 256 KB block   %6.2f TB/s chip   %5.1f B/ns/CU\n", name, NW, CONTIG ? "1 KB contiguous" : "8 rows x 128 B ",
           ms * 1e3 / passes, bytes / (ms * 1e-3) / 1e12, 262144.0 * passes / (ms * 1e6));
}

int main()
{
    char *a; uint32_t *sink;
    hipMalloc(&a, 262144); hipMemset(a, 1, 262144); hipMalloc(&sink, 4096);
#define ALL(MODE, NAME) run<MODE, 4, 0>(a, sink, NAME); run<MODE, 8, 0>(a, sink, NAME); run<MODE, 16, 0>(a, sink, NAME); run<MODE, 8, 1>(a, sink, NAME);
    ALL(0, "LDS-DMA global_load_lds_dwordx4")
    ALL(1, "global_load_dwordx4 -> VGPR")
    ALL(2, "global_load_dwordx4 -> VGPR -> ds_write_b128")
    return 0;
}
