"""In-kernel phase timeline of the skinny 1-bit GEMM (2 <= T <= 64), stamps build (-DOB_PROFILE_STAMPS, OB_TIMING=1).
Usage: OB_LIB=onebit_amd/csrc/libob_stamps.so python tools/skinny_phase_probe.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onebit_amd import _lib
_lib.LIB_PATH = os.environ["OB_LIB"]
os.environ["OB_TIMING"] = "1"
from onebit_amd import BitLinearInf
lib = _lib.load()
lib.onebit_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
names = ["entry", "loads of phase 0 issued", "phase 0 in LDS (barrier)", "MFMAs of phase 0 issued", "end of phase 0", "end of phase 1",
         "end of phase 3", "end of K loop", "partials in LDS (barrier)", "end"]
g = torch.Generator().manual_seed(3)
for (T, K, N) in ((32, 4096, 4096), (32, 4096, 11008), (32, 11008, 4096), (16, 4096, 11008), (64, 4096, 11008)):
    # several layers' weights so the measured launch streams from HBM, not the Infinity Cache
    mods = []
    for i in range(24):
        m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
        m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
        m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
        m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
        m.layernorm = torch.nn.Identity()
        mods.append(m)
    x = torch.randn(T, K, generator=g).half().to(dev)
    for m in mods: m(x)
    torch.cuda.synchronize()
    nb = min((N + 63) // 64, 512)
    acc = []
    for rep in range(6):
        for m in mods[:-1]: m(x)
        torch.cuda.synchronize()
        mods[-1](x)
        buf = np.zeros((512, 8, 16), dtype=np.uint64)
        lib.onebit_debug_read_timing(buf.ctypes.data, 256)
        t = buf[:nb].astype(np.float64)
        t[t == 0] = np.nan
        acc.append(t - t[:, :, :1])                # per wave, cycles since its own entry
    a = np.concatenate(acc, axis=0)
    print("T=%d K=%d N=%d (%d workgroups): cycles since the wave's entry   min / median / max over waves" % (T, K, N, nb))
    with np.errstate(all="ignore"):
        lo, med, hi = np.nanmin(a, axis=(0, 1)), np.nanmedian(a, axis=(0, 1)), np.nanmax(a, axis=(0, 1))
    for i, nm in enumerate(names):
        print("    %-30s %7.0f %7.0f %7.0f" % (nm, lo[i], med[i], hi[i]))
    del mods
