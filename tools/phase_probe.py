"""In-kernel phase timeline of the fused decode GEMV launches (stamps-only profiling build -DOB_PROFILE_STAMPS: the
ablation build's early-return branches put vmcnt(0) behind every weight load and stretch the head).  Stamps are kept in registers and written at kernel end.  Usage:
  [OB_EXTRA="-DOB_STRIDED_LOADS"] [OB_PROBE_STATS=0|1] python tools/phase_probe.py"""
import ctypes, os, sys, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("OB_LIB") or "/tmp/libonebit_prof_%d.so" % os.getpid()
if not os.environ.get("OB_LIB"):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                       "-Wno-unused-value", "-DOB_PROFILE_STAMPS", *os.environ.get("OB_EXTRA", "").split(), "-o", so,
                       os.path.join(ROOT, "onebit_amd/csrc/onebit_hip.hip"), os.path.join(ROOT, "onebit_amd/csrc/onebit_mixed.hip")])
from onebit_amd import _lib
_lib.LIB_PATH = so
os.environ["OB_TIMING"] = "1"
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import fused_gemv, tile_stats_floats, PRO_PLAIN, PRO_RES_LN_RMS, PRO_SWIGLU
USE_STATS = os.environ.get("OB_PROBE_STATS", "1") == "1"
lib = _lib.load()
dev = torch.device("cuda:0")
NL = 4
cfg = OneBitLlamaConfig(num_hidden_layers=NL)
model = build_synthetic_model(cfg, seed=1, device=dev)
H, I = cfg.hidden_size, cfg.intermediate_size
f16 = torch.float16
hres, uprev, hout = (torch.randn(H, device=dev).to(f16) for _ in range(3))
ug, uu = torch.randn(I, device=dev).to(f16), torch.randn(I, device=dev).to(f16)
oq, ok, ov, oo, og, ou, od = (torch.empty(n, device=dev, dtype=f16) for n in (H, H, H, H, I, I, H))
stH, stH2, stI, stI2, stq, stk, stv = (torch.zeros(tile_stats_floats(n), device=dev) for n in (H, H, I, I, H, H, H))
so_ = lambda *a: list(a) if USE_STATS else None
si = lambda **kw: kw if USE_STATS else {}
def run(kind, l):
    if kind == "o": fused_gemv([l.self_attn.o_proj], [oo], PRO_PLAIN, xin=hres, stats_out=so_(stH2))
    elif kind == "qkv": fused_gemv([l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj], [oq, ok, ov], PRO_RES_LN_RMS, hres_in=hres, u_prev=uprev, hres_out=hout, rms_w=l.input_layernorm.weight, stats_out=so_(stq, stk, stv), **si(st_prev=stH))
    elif kind == "gateup": fused_gemv([l.mlp.gate_proj, l.mlp.up_proj], [og, ou], PRO_RES_LN_RMS, hres_in=hres, u_prev=uprev, hres_out=hout, rms_w=l.post_attention_layernorm.weight, stats_out=so_(stI, stI2), **si(st_prev=stH))
    else: fused_gemv([l.mlp.down_proj], [od], PRO_SWIGLU, u_gate=ug, u_up=uu, stats_out=so_(stH2), **si(st_gate=stI, st_up=stI2))
lib.onebit_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["entry", "loads issued", "LN stats ready", "RMS reduced", "x ready", "amax", "digits in LDS / flag seen", "first MFMA group", "MFMA done", "after partials barrier", "end"]
print("build:", os.environ.get("OB_LIB", os.environ.get("OB_EXTRA", "(default)")), " stats:", USE_STATS, " OB_DEC2:", os.environ.get("OB_DEC2", "0"))
layers = list(model.model.layers)
def report(tag, a):
    with np.errstate(all="ignore"):
        lo, med, hi = np.nanmin(a, axis=(0, 1)), np.nanmedian(a, axis=(0, 1)), np.nanmax(a, axis=(0, 1))
    print("  ", tag)
    for i, nm in enumerate(names):
        if not np.isnan(med[i]):
            print("    %-26s %7.0f %7.0f %7.0f" % (nm, lo[i], med[i], hi[i]))
for kind in ("o", "qkv", "gateup", "down"):
    for rep in range(3):
        for l in layers: run(kind, l)
    torch.cuda.synchronize()
    acc = []
    for rep in range(8):
        for l in layers[:-1]: run(kind, l)          # the measured launch follows other launches, as in a decode chain
        torch.cuda.synchronize()
        buf = np.zeros((256, 16, 16), dtype=np.uint64)
        run(kind, layers[-1])
        lib.onebit_debug_read_timing(buf.ctypes.data, 256)
        t = buf.astype(np.float64)
        t[t == 0] = np.nan
        acc.append((t - t[:, :, :1])[:, :, :11])     # per wave: cycles since ITS OWN entry (XCD clocks are not synchronised)
    with np.errstate(all="ignore"):
        a = np.nanmedian(np.stack(acc), axis=0)       # [256 wg][16 waves][11]
    print(kind, "-- cycles since the wave entered (min / median / max over the grid's waves)")
    dec2 = int(os.environ.get("OB_DEC2", "0"))
    if not (dec2 == 2 or (dec2 == 1 and kind != "o")):   # single-role kernel: dbg holds [wg][8 waves][16]
        b = buf.reshape(-1)[: 256 * 8 * 16].reshape(256, 8, 16).astype(np.float64)
        b[b == 0] = np.nan
        report("all waves", (b - b[:, :, :1])[:, :, :11])
    else:
        report("prologue waves", a[:, :8])
        report("matrix waves", a[:, 8:])
