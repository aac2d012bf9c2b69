"""What would a split-K form of down_proj cost?  (round 5, after profiles/r05_decode_phase_timeline.txt: the launch is bound by
the VALU work of its redundant SwiGLU prologue -- every workgroup evaluates all 11008 elements for 16 output rows.)
Proposed: 2 K-halves x 128 workgroups of TWO tiles, fp32 partial sums, the consumer adds them.  Proxies timed here as graph chains
over 16 distinct weight sets (HIP events):
  full     down as today            K = 11008, N = 4096 (256 workgroups x 1 tile)
  halfK    half the prologue        K =  5632, N = 4096 (256 x 1 tile: half the weight bytes too)
  halfK2   half prologue, 2 tiles   K =  5632, N = 8192 (256 x 2 tiles: the proposed per-workgroup work)
  qkv / qkv-nopst   the consumer with / without the producer's tile partials (what it loses when it has to add two partial vectors)
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf
from onebit_amd.engine import fused_gemv, tile_stats_floats, PRO_RES_LN_RMS, PRO_SWIGLU
dev = torch.device("cuda:0")
f16 = torch.float16
NSET = 16

def mk(K, N, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    m = BitLinearInf(K, N, dtype=f16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    return m

def time_chain(fn, n):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / (20 * n))
    return float(np.median(best))

def down_case(K, N):
    mods = [mk(K, N, 100 + i) for i in range(NSET)]
    ug, uu = torch.randn(K, device=dev).to(f16), torch.randn(K, device=dev).to(f16)
    stg, stu, sto = (torch.zeros(tile_stats_floats(n), device=dev) for n in (K, K, N))
    # plausible partials so that the statistics are finite
    for v, s in ((ug, stg), (uu, stu)):
        t = v.float().view(-1, 16)
        s[: 2 * t.shape[0]].view(-1, 2)[:, 0] = t.sum(1)
        s[: 2 * t.shape[0]].view(-1, 2)[:, 1] = ((t - t.mean(1, keepdim=True)) ** 2).sum(1)
    out = torch.empty(N, device=dev, dtype=f16)
    def fn():
        for m in mods:
            fused_gemv([m], [out], PRO_SWIGLU, u_gate=ug, u_up=uu, stats_out=[sto], st_gate=stg, st_up=stu)
    return time_chain(fn, NSET)

def qkv_case(pst):
    H = 4096
    mods = [[mk(H, H, 200 + 3 * i + j) for j in range(3)] for i in range(NSET)]
    hres, uprev, hout = (torch.randn(H, device=dev).to(f16) for _ in range(3))
    w = torch.ones(H, device=dev, dtype=f16)
    st = torch.zeros(tile_stats_floats(H), device=dev)
    t = uprev.float().view(-1, 16)
    st[: 2 * t.shape[0]].view(-1, 2)[:, 0] = t.sum(1)
    st[: 2 * t.shape[0]].view(-1, 2)[:, 1] = ((t - t.mean(1, keepdim=True)) ** 2).sum(1)
    outs = [torch.empty(H, device=dev, dtype=f16) for _ in range(3)]
    so = [torch.zeros(tile_stats_floats(H), device=dev) for _ in range(3)]
    kw = dict(st_prev=st) if pst else {}
    def fn():
        for m3 in mods:
            fused_gemv(m3, outs, PRO_RES_LN_RMS, hres_in=hres, u_prev=uprev, hres_out=hout, rms_w=w, stats_out=so, **kw)
    return time_chain(fn, NSET)

print("down full   K=11008 N=4096 : %.2f us per launch" % down_case(11008, 4096))
print("down halfK  K= 5632 N=4096 : %.2f us per launch" % down_case(5632, 4096))
print("down halfK2 K= 5632 N=8192 : %.2f us per launch  (the proposed per-workgroup work: half the prologue, two tiles)" % down_case(5632, 8192))
print("down qtrK4  K= 2816 N=16384: %.2f us per launch  (quarter prologue, four tiles)" % down_case(2816, 16384))
print("qkv with producer partials : %.2f us per launch" % qkv_case(True))
print("qkv recomputing statistics : %.2f us per launch" % qkv_case(False))
