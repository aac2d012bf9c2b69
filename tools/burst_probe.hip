// Burst-load probe: what bounds the load phase of a decode GEMV launch?  256 workgroups x 512
// threads; every workgroup reads NV shared 8 KB vectors (the same lines in every workgroup: the
// redundant prologue inputs) and NW 16-byte words per lane of its own packed-weight tiles (unique
// lines, streamed once).  Reports us / launch in a graph chain over distinct weights and the
// in-kernel cycle at which a wave has issued / received everything.
//   weights layout 0: slot s of workgroup b = tile s * G + b   (2 MB apart: one page per slot)
//                  1: slot s of workgroup b = tile b * NW + s  (contiguous per workgroup)
//   vectors  layout 0: each vector in its own 2 MB-aligned allocation; 1: packed back to back
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Args { const u32x4 *vec[8]; const u32x4 *w; unsigned long long *dbg; u32x4 *sink; long long wofs; int wlayout; int rot; };

template <int NV, int NW, int POL>
__global__ __launch_bounds__(512) void k_burst(const Args a)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, G = gridDim.x;
    u32x4 v[NV > 0 ? NV : 1], w[NW > 0 ? NW : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = a.rot ? ((tid + 64 * b) & 511) : tid;       // rot: workgroups start at different lines
        if (POL == 1) v[i] = __builtin_nontemporal_load(a.vec[i] + idx);
        else v[i] = a.vec[i][idx];
    }
#pragma unroll
    for (int s = 0; s < NW; ++s) {
        const long long tile = a.wlayout ? (long long)b * NW + s : (long long)s * G + b;
        const u32x4 *p = a.w + a.wofs + tile * 512 + (lane & 15) * 32 + wave * 4 + (lane >> 4);   // 16 rows x 512 B
        w[s] = __builtin_nontemporal_load(p);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NV; ++i) acc ^= v[i];
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t2 = 0;
    if (NV > 0) { asm volatile("s_nop 0" :: "v"(acc)); t2 = __builtin_readcyclecounter(); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NW; ++s) acc ^= w[s];
    asm volatile("s_nop 0" :: "v"(acc));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t3 = __builtin_readcyclecounter();
    if (acc[0] == 0x12345u && acc[1] == 0x777u) a.sink[tid] = acc;
    if (a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + (size_t)(b * 8 + wave) * 4;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
    }
}

// rolling issue: groups of GRP loads, at most WIN groups in flight (wait for the oldest group before
// issuing the next) -- the schedule a wave can follow without ever blocking in the issue of a load
template <int NV, int NW, int GRP, int WIN, int THREADS>
__global__ __launch_bounds__(THREADS) void k_roll(const Args a)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    constexpr int WAVES = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    u32x4 v[NV > 0 ? NV : 1], w[NW];
    if (tid < 512) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = a.vec[i][tid];
    }
    u32x4 acc = {0, 0, 0, 0};
    constexpr int NG = (NW + GRP - 1) / GRP;
    unsigned long long t1 = 0;
#pragma unroll
    for (int g = 0; g < NG + WIN; ++g) {
        if (g < NG) {
#pragma unroll
            for (int s = g * GRP; s < (g + 1) * GRP && s < NW; ++s) {
                // NW * 8 chunks of 1 KB per workgroup dealt to WAVES waves
                const long long tile = (long long)b * NW + s;
                const int chunk = wave % 8, part = wave / 8;          // 16 waves: two waves share a chunk (half the rows each)
                const u32x4 *p = a.w + a.wofs + tile * 512 + ((lane & 15) * 32 + chunk * 4 + (lane >> 4));
                (void)part;
                w[s] = __builtin_nontemporal_load(p);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g == WIN - 1 || (NG < WIN && g == NG - 1)) t1 = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g >= WIN) {
#pragma unroll
            for (int s = (g - WIN) * GRP; s < (g - WIN + 1) * GRP && s < NW; ++s) acc ^= w[s];
            asm volatile("s_nop 0" :: "v"(acc));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) acc ^= v[i];
    asm volatile("s_nop 0" :: "v"(acc));
    const unsigned long long t3 = __builtin_readcyclecounter();
    if (acc[0] == 0x12345u && acc[1] == 0x777u) a.sink[tid] = acc;
    if (a.dbg && lane == 0 && wave < 8) {
        unsigned long long *d = a.dbg + (size_t)(b * 8 + wave) * 4;
        d[0] = t0; d[1] = t1; d[2] = t0; d[3] = t3;
    }
}

template <typename F> static float graph_chain(F launch, int n, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / (10 * n);
}

static Args g_a;
static unsigned long long *g_dbg;
static hipStream_t g_s;

template <int NV, int NW, int POL>
static void run(const char *tag, int wlayout, int vlayout, int rot, const u32x4 *const *vsep, const u32x4 *const *vpack)
{
    Args a = g_a;
    for (int i = 0; i < 8; ++i) a.vec[i] = vlayout ? vpack[i] : vsep[i];
    a.wlayout = wlayout; a.rot = rot; a.dbg = nullptr;
    const long long per = (long long)256 * (NW > 0 ? NW : 1) * 512;              // u32x4 per launch
    auto f = [&](int i) { Args b = a; b.wofs = (long long)(i % 32) * per; hipLaunchKernelGGL((k_burst<NV, NW, POL>), dim3(256), dim3(512), 0, g_s, b); };
    const float us = graph_chain(f, 64, g_s);
    // in-kernel timeline of one launch that follows other launches
    Args b = a; b.dbg = g_dbg; b.wofs = 33 * per;
    for (int i = 0; i < 4; ++i) f(i);
    hipLaunchKernelGGL((k_burst<NV, NW, POL>), dim3(256), dim3(512), 0, g_s, b);
    hipStreamSynchronize(g_s);
    std::vector<unsigned long long> h(256 * 8 * 4);
    hipMemcpy(h.data(), g_dbg, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> d1, d2, d3;
    for (int i = 0; i < 256 * 8; ++i) {
        d1.push_back((double)(h[i * 4 + 1] - h[i * 4]));
        if (NV > 0) d2.push_back((double)(h[i * 4 + 2] - h[i * 4]));
        d3.push_back((double)(h[i * 4 + 3] - h[i * 4]));
    }
    auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> &v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
    printf("%-34s NV=%d NW=%d wl=%d vl=%d rot=%d pol=%d : %6.2f us/launch | wave cycles med/max: issued %5.0f/%5.0f  vectors %5.0f/%5.0f  all %5.0f/%5.0f | %.1f KB/CU\n",
           tag, NV, NW, wlayout, vlayout, rot, POL, us, med(d1), mx(d1), med(d2), mx(d2), med(d3), mx(d3), (NV * 8192 + NW * 8192) / 1024.0);
}

template <int NV, int NW, int GRP, int WIN, int THREADS>
static void run_roll(const char *tag, const u32x4 *const *vpack)
{
    Args a = g_a;
    for (int i = 0; i < 8; ++i) a.vec[i] = vpack[i];
    a.dbg = nullptr;
    const long long per = (long long)256 * NW * 512;
    auto f = [&](int i) { Args b = a; b.wofs = (long long)(i % 32) * per; hipLaunchKernelGGL((k_roll<NV, NW, GRP, WIN, THREADS>), dim3(256), dim3(THREADS), 0, g_s, b); };
    const float us = graph_chain(f, 64, g_s);
    Args b = a; b.dbg = g_dbg; b.wofs = 33 * per;
    for (int i = 0; i < 4; ++i) f(i);
    hipLaunchKernelGGL((k_roll<NV, NW, GRP, WIN, THREADS>), dim3(256), dim3(THREADS), 0, g_s, b);
    hipStreamSynchronize(g_s);
    std::vector<unsigned long long> h(256 * 8 * 4);
    hipMemcpy(h.data(), g_dbg, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> d1, d3;
    for (int i = 0; i < 256 * 8; ++i) { d1.push_back((double)(h[i * 4 + 1] - h[i * 4])); d3.push_back((double)(h[i * 4 + 3] - h[i * 4])); }
    std::sort(d1.begin(), d1.end()); std::sort(d3.begin(), d3.end());
    printf("%-26s NV=%d NW=%d grp=%d win=%d threads=%4d : %6.2f us/launch | first window issued med/max %5.0f/%5.0f  all arrived %5.0f/%5.0f\n",
           tag, NV, NW, GRP, WIN, THREADS, us, d1[d1.size() / 2], d1.back(), d3[d3.size() / 2], d3.back());
}

int main()
{
    hipStreamCreate(&g_s);
    const u32x4 *vsep[8], *vpack[8];
    for (int i = 0; i < 8; ++i) { void *p; hipMalloc(&p, 2 << 20); hipMemset(p, 1, 2 << 20); vsep[i] = (const u32x4 *)p; }
    { char *p; hipMalloc((void **)&p, 2 << 20); hipMemset(p, 1, 2 << 20); for (int i = 0; i < 8; ++i) vpack[i] = (const u32x4 *)(p + i * 8192); }
    void *w; const size_t wbytes = (size_t)36 * 256 * 6 * 8192; hipMalloc(&w, wbytes); hipMemset(w, 3, wbytes);
    hipMalloc((void **)&g_dbg, 256 * 8 * 4 * 8);
    void *sink; hipMalloc(&sink, 1 << 16);
    g_a.w = (const u32x4 *)w; g_a.sink = (u32x4 *)sink; g_a.wofs = 0;
    run<0, 0, 0>("empty", 0, 0, 0, vsep, vpack);
    run<5, 6, 0>("gateup-like contiguous, packed", 1, 1, 0, vsep, vpack);
    run<4, 2, 0>("head", 1, 1, 0, vsep, vpack);
    run<5, 2, 0>("head", 1, 1, 0, vsep, vpack);
    run<6, 2, 0>("head", 1, 1, 0, vsep, vpack);
    run<7, 2, 0>("head", 1, 1, 0, vsep, vpack);
    run<8, 2, 0>("head", 1, 1, 0, vsep, vpack);
    run<3, 2, 0>("head", 1, 1, 0, vsep, vpack);
    run<2, 2, 0>("head", 1, 1, 0, vsep, vpack);
    run<8, 0, 0>("vectors", 1, 1, 0, vsep, vpack);
    run<6, 0, 0>("vectors", 1, 1, 0, vsep, vpack);
    run<7, 0, 0>("vectors", 1, 1, 0, vsep, vpack);
    // how many loads can a wave / a CU have in flight before the issue blocks?
    run_roll<0, 6, 1, 1, 512>("roll", vpack);
    run_roll<0, 6, 1, 2, 512>("roll", vpack);
    run_roll<0, 6, 1, 3, 512>("roll", vpack);
    run_roll<0, 6, 2, 1, 512>("roll", vpack);
    run_roll<0, 6, 2, 2, 512>("roll", vpack);
    run_roll<0, 6, 3, 1, 512>("roll", vpack);
    run_roll<0, 6, 3, 2, 512>("roll", vpack);
    run_roll<0, 6, 6, 1, 512>("roll (all at once)", vpack);
    run_roll<5, 6, 2, 1, 512>("roll + vectors", vpack);
    run_roll<5, 6, 2, 2, 512>("roll + vectors", vpack);
    run_roll<0, 3, 3, 1, 256>("4 waves, 3 loads at once", vpack);
    run_roll<0, 6, 6, 1, 256>("4 waves, 6 loads at once", vpack);
    run_roll<0, 3, 3, 1, 1024>("16 waves, 3 loads at once", vpack);
    run_roll<0, 2, 2, 1, 1024>("16 waves, 2 loads at once", vpack);
    return 0;
}
