"""Per-launch wall time of the fused decode GEMV launches (graph replay, HIP events), per OB_ABLATE mode.
Usage: OB_ABLATE=m python tools/gemv_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("OB_PROFILE_BUILD"):     # profiling build (-DOB_PROFILE_ABLATE: OB_ABLATE=4 / 10+i return at stamp i)
    import subprocess
    from onebit_amd import _lib
    so = "/tmp/libonebit_prof.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
                           "-DOB_PROFILE_ABLATE", *os.environ.get("OB_EXTRA", "").split(), "-o", so,
                           os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "onebit_amd/csrc/onebit_hip.hip"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "onebit_amd/csrc/onebit_mixed.hip")])
    _lib.LIB_PATH = so
if os.environ.get("OB_LIB"):
    from onebit_amd import _lib
    _lib.LIB_PATH = os.environ["OB_LIB"]
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import fused_gemv, tile_stats_floats, PRO_PLAIN, PRO_RES_LN_RMS, PRO_SWIGLU
USE_STATS = os.environ.get("OB_PROBE_STATS", "1") == "1"      # producers' tile partials (the decode_step path)

dev = torch.device("cuda:0")
cfg = OneBitLlamaConfig(num_hidden_layers=int(os.environ.get("OB_PROBE_LAYERS", "8")))
model = build_synthetic_model(cfg, seed=1, device=dev)
H, I = cfg.hidden_size, cfg.intermediate_size
f16 = torch.float16
hres, uprev, hout = (torch.randn(H, device=dev).to(f16) for _ in range(3))
ug, uu = torch.randn(I, device=dev).to(f16), torch.randn(I, device=dev).to(f16)
oq, ok, ov, oo, og, ou, od = (torch.empty(n, device=dev, dtype=f16) for n in (H, H, H, H, I, I, H))
L = list(model.model.layers)
stH, stI, stI2 = (torch.zeros(tile_stats_floats(n), device=dev) for n in (H, I, I))
so = lambda *a: list(a) if USE_STATS else None
si = lambda **kw: kw if USE_STATS else {}

def chain(kind):
    for l in L:
        if kind == "o":
            fused_gemv([l.self_attn.o_proj], [oo], PRO_PLAIN, xin=hres, stats_out=so(stH))
        elif kind == "qkv":
            fused_gemv([l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj], [oq, ok, ov], PRO_RES_LN_RMS,
                       hres_in=hres, u_prev=uprev, hres_out=hout, rms_w=l.input_layernorm.weight, stats_out=so(stH, stH, stH), **si(st_prev=stH))
        elif kind == "gateup":
            fused_gemv([l.mlp.gate_proj, l.mlp.up_proj], [og, ou], PRO_RES_LN_RMS,
                       hres_in=hres, u_prev=uprev, hres_out=hout, rms_w=l.post_attention_layernorm.weight, stats_out=so(stI, stI2), **si(st_prev=stH))
        else:
            fused_gemv([l.mlp.down_proj], [od], PRO_SWIGLU, u_gate=ug, u_up=uu, stats_out=so(stH), **si(st_gate=stI, st_up=stI2))

out = []
for kind in ("o", "qkv", "gateup", "down"):
    chain(kind); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(kind)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    out.append("%s %.2f" % (kind, e0.elapsed_time(e1) * 1e3 / (20 * len(L))))
print("OB_ABLATE=%s stats=%d us/launch:" % (os.environ.get("OB_ABLATE", "0"), USE_STATS), "  ".join(out))
