// Probe: semantics of ds_read_b64_tr_b16 (gfx950 LDS transpose read).  LDS holds lds[i] = i (16-bit elements); every lane
// supplies a byte address; the result's 4 elements per lane are printed for three address patterns.
//   hipcc -O2 --offload-arch=gfx950 tools/tr16_probe.hip -o tools/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out, const int *addr)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) v4s lv4s;
    lv4s *p = (lv4s *)((__attribute__((address_space(3))) char *)lds + addr[threadIdx.x]);
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main()
{
    int h[64]; unsigned short o[256];
    int *d; unsigned short *od;
    hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
    for (int pat = 0; pat < 4; ++pat) {
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, i = l & 15;
            if (pat == 0) h[l] = 8 * l;                                        // lane l: chunk l (4 consecutive elements)
            else if (pat == 1) h[l] = 0;                                       // uniform
            else if (pat == 2) h[l] = ((i >> 2) * 16 + 4 * (i & 3)) * 2 + g * 128;   // 4 x 16 row-major block per group, contiguous
            else h[l] = ((i >> 2) * 136 + 4 * (i & 3)) * 2 + g * (4 * 136 * 2);        // rows 136 elements apart, groups 4 rows apart
        }
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, od, d);
        hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d (addr %5d B = elem %4d): %5u %5u %5u %5u\n", l, h[l], h[l] / 2, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
            if (l == 19 && pat != 0) { printf("  ...\n"); l = 47; }
        }
    }
    return 0;
}
