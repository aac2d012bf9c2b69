"""In-kernel phase timing of the fused decode GEMV (profiling build: -DOB_PROFILE_ABLATE, OB_TIMING=1)."""
import ctypes, os, sys, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libonebit_prof.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                       "-Wno-unused-value", "-DOB_PROFILE_ABLATE", "-o", so, os.path.join(ROOT, "onebit_amd/csrc/onebit_hip.hip")])
from onebit_amd import _lib
_lib.LIB_PATH = so
os.environ["OB_TIMING"] = "1"
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import fused_gemv, PRO_PLAIN, PRO_RES_LN_RMS, PRO_SWIGLU
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = OneBitLlamaConfig(num_hidden_layers=1)
model = build_synthetic_model(cfg, seed=1, device=dev)
H, I = cfg.hidden_size, cfg.intermediate_size
f16 = torch.float16
hres, uprev, hout = (torch.randn(H, device=dev).to(f16) for _ in range(3))
ug, uu = torch.randn(I, device=dev).to(f16), torch.randn(I, device=dev).to(f16)
oq, ok, ov, oo, og, ou, od = (torch.empty(n, device=dev, dtype=f16) for n in (H, H, H, H, I, I, H))
l = model.model.layers[0]
def run(kind):
    if kind == "o": fused_gemv([l.self_attn.o_proj], [oo], PRO_PLAIN, xin=hres)
    elif kind == "qkv": fused_gemv([l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj], [oq, ok, ov], PRO_RES_LN_RMS, hres_in=hres, u_prev=uprev, hres_out=hout, rms_w=l.input_layernorm.weight)
    elif kind == "gateup": fused_gemv([l.mlp.gate_proj, l.mlp.up_proj], [og, ou], PRO_RES_LN_RMS, hres_in=hres, u_prev=uprev, hres_out=hout, rms_w=l.post_attention_layernorm.weight)
    else: fused_gemv([l.mlp.down_proj], [od], PRO_SWIGLU, u_gate=ug, u_up=uu)
lib.onebit_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["entry->loads issued", "->prologue math done", "->amax reduced", "->digits in LDS (barrier)", "->MFMA done", "->end"]
idx = [0, 1, 2, 3, 4, 5, 7]
for kind in ("o", "qkv", "gateup", "down"):
    for _ in range(3): run(kind)
    torch.cuda.synchronize()
    buf = np.zeros((256, 8, 8), dtype=np.uint64)
    lo = np.zeros(len(idx)); hi = np.zeros(len(idx)); n = 0
    for rep in range(10):
        run(kind)
        lib.onebit_debug_read_timing(buf.ctypes.data, 256)
        t = buf.astype(np.float64)
        for b in range(0, 256, 8):
            st = t[b][:, idx]                       # [8 waves][stamps]
            if (st == 0).any(): st = np.where(st == 0, np.nan, st)
            base = np.nanmin(st[:, 0])
            lo += np.nanmin(st, axis=0) - base; hi += np.nanmax(st, axis=0) - base; n += 1
    lo /= n; hi /= n
    print(os.environ.get("OB_DECODE_MATH", "i8"), kind.ljust(7), "cycles since first wave entry, earliest..latest wave:",
          "  ".join("%s %d..%d" % (nm, a, b) for nm, a, b in zip(["entry"] + names, np.nan_to_num(lo, nan=-1), np.nan_to_num(hi, nan=-1))))
