"""Per-launch time of the skinny 1-bit GEMM forms (ob_skinny.h on x with the scaling fused; the pre-scaled route on
producer-scaled rows: ob_skinny3.h) at the LLaMA-7B projection shapes, T tokens, one projection per launch, graph-replayed chains
over 32 distinct weight sets (like bench.py's roofline chains).  Usage: python tools/skinny_cmp.py [T]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(0)
def mk(K, N):
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    return m
def chain_us(fn, mods):
    for m in mods: fn(m)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for m in mods: fn(m)
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * len(mods))
for name, K, N in (("q/o 4096->4096", 4096, 4096), ("gate 4096->11008", 4096, 11008), ("down 11008->4096", 11008, 4096)):
    mods = [mk(K, N) for _ in range(24)]
    if os.environ.get("REUSE"): mods = [mods[0]] * 24        # the same packed matrix every launch: its rows stay in the XCDs' L2
    x = torch.randn(T, K, generator=g).half().to(dev)
    a = x * mods[0].input_factor.data
    ok = mods[0].prescaled_ok(T)
    t1 = chain_us(lambda m: m.pre_layernorm(x), mods)
    t2 = chain_us(lambda m: m.pre_layernorm_prescaled(a), mods) if ok else float("nan")
    u1 = mods[0].pre_layernorm(x).float()
    u2 = mods[0].pre_layernorm_prescaled(a).float() if ok else u1
    # fp32 reference of the same op: fp16(fp16(sign(W) . a) * g)
    wb = mods[0].weight.data.view(torch.uint8)
    sg = 1.0 - 2.0 * ((wb.unsqueeze(-1) >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(N, K).float()
    ref = ((a.float() @ sg.t()).half().float() * mods[0].weight_scale.data.float()).half().float()
    print("T=%d %-18s form 1 %6.2f us   pre-scaled %6.2f us   max |form1 - prescaled| %.3g   max |prescaled - fp32 ref| %.3g  (max |ref| %.3g)"
          % (T, name, t1, t2, (u1 - u2).abs().max().item(), (u2 - ref).abs().max().item(), ref.abs().max().item()), flush=True)
    del mods
