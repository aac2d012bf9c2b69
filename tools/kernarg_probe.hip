// Probe (round 4): what the kernarg fetch at the head of a decode launch costs when the caches are cold, and whether
// kernarg preload (scalar leading arguments + -mllvm -amdgpu-kernarg-preload-count=16) removes it.
// Every kernel: 256 x 512 threads; each lane issues one 16-byte load from a pointer held in the kernarg (the
// "prologue vector"), then streams `mb` MB of packed-weight-like data (non-temporal, one distinct buffer per launch of
// the chain, 64 buffers: > Infinity Cache) so that consecutive launches find neither L2 nor MALL warm, and stores a sum.
// Build twice:  hipcc -O3 --offload-arch=gfx950 tools/kernarg_probe.hip -o tools/kernarg_probe
//               hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_probe.hip -o tools/kernarg_probe_pl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Args { const u32x4 *vec; const u32x4 *w; u32x4 *out; int per_lane; int pad; };
template <bool STREAM>
__device__ __forceinline__ void body(const u32x4 *vec, const u32x4 *w, u32x4 *out, int per_lane)
{
    const int tid = threadIdx.x + blockIdx.x * 512;
    u32x4 acc = vec[threadIdx.x & 511];
    if (STREAM) {
        for (int i = 0; i < per_lane; ++i) {
            const u32x4 v = __builtin_nontemporal_load(w + (size_t)i * 256 * 512 + tid);
            acc ^= v;
        }
    }
    if (acc[0] == 0x12345u) out[tid] = acc;
}
template <bool STREAM> __global__ __launch_bounds__(512) void k_struct(const Args a) { body<STREAM>(a.vec, a.w, a.out, a.per_lane); }
template <bool STREAM> __global__ __launch_bounds__(512) void k_scalar(const u32x4 *vec, const u32x4 *w, u32x4 *out, int per_lane) { body<STREAM>(vec, w, out, per_lane); }
__global__ __launch_bounds__(512) void k_empty() {}

template <typename F> static float graph_chain(F launch, int n, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / (10 * n);
}
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    const int NB = 64;
    const size_t mb = 6, bytes = mb << 20;                  // 6 MB per launch: the q|k|v / down launch
    const int per_lane = (int)(bytes / (256 * 512 * 16));    // 16-byte loads per lane
    std::vector<u32x4 *> w(NB), v(NB);
    u32x4 *out; hipMalloc(&out, 256 * 512 * 16);
    for (int i = 0; i < NB; ++i) { hipMalloc(&w[i], bytes); hipMemset(w[i], 1, bytes); hipMalloc(&v[i], 8192); hipMemset(v[i], 2, 8192); }
    printf("per-lane loads %d, %zu MB per launch, %d buffers\n", per_lane, mb, NB);
    printf("empty                    : %6.2f us/launch\n", graph_chain([&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s); }, 256, s));
    for (int rep = 0; rep < 2; ++rep) {
        printf("struct,  vector only     : %6.2f us/launch\n", graph_chain([&](int i) { Args a = {v[i % NB], w[i % NB], out, per_lane, 0}; hipLaunchKernelGGL(k_struct<false>, dim3(256), dim3(512), 0, s, a); }, 256, s));
        printf("scalars, vector only     : %6.2f us/launch\n", graph_chain([&](int i) { hipLaunchKernelGGL(k_scalar<false>, dim3(256), dim3(512), 0, s, (const u32x4 *)v[i % NB], (const u32x4 *)w[i % NB], out, per_lane); }, 256, s));
        printf("struct,  vector + stream : %6.2f us/launch\n", graph_chain([&](int i) { Args a = {v[i % NB], w[i % NB], out, per_lane, 0}; hipLaunchKernelGGL(k_struct<true>, dim3(256), dim3(512), 0, s, a); }, 256, s));
        printf("scalars, vector + stream : %6.2f us/launch\n", graph_chain([&](int i) { hipLaunchKernelGGL(k_scalar<true>, dim3(256), dim3(512), 0, s, (const u32x4 *)v[i % NB], (const u32x4 *)w[i % NB], out, per_lane); }, 256, s));
    }
    return 0;
}
