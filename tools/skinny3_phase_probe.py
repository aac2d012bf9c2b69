"""In-kernel phase timeline of the LDS-DMA skinny 1-bit GEMM (ob_skinny3.h, pre-scaled rows, 2 <= T <= 64), stamps build
(-DOB_PROFILE_STAMPS, OB_TIMING=1).
Usage: OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so python tools/skinny3_phase_probe.py"""
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
names = ["entry", "first pieces requested", "piece 0 landed", "MFMAs of piece 0 issued", "end of piece 0", "end of piece 1",
         "end of piece 3", "end of K loop", "partials in LDS (barriers)", "end"]
g = torch.Generator().manual_seed(3)
for (T, K, N) in ((32, 4096, 4096), (32, 4096, 12288), (32, 4096, 22016), (32, 11008, 4096)):
    mods = []
    for i in range(12):       # several layers' weights so the measured launch streams from HBM, not the Infinity Cache
        m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
        m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
        m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
        m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
        mods.append(m)
    a = torch.randn(T, K, generator=g).half().to(dev)
    for m in mods: m.pre_layernorm_prescaled(a)
    torch.cuda.synchronize()
    acc = []
    for rep in range(6):
        for m in mods[:-1]: m.pre_layernorm_prescaled(a)
        torch.cuda.synchronize()
        mods[-1].pre_layernorm_prescaled(a)
        buf = np.zeros((512, 8, 16), dtype=np.uint64)
        lib.onebit_debug_read_timing(buf.ctypes.data, 256)
        t = buf.astype(np.float64)
        t[t == 0] = np.nan
        t = t[~np.isnan(t[:, 0, 0])]
        acc.append(t - t[:, :, :1])                # per wave, cycles since its own entry
    s = np.concatenate(acc, axis=0)
    print("T=%d K=%d N=%d (%d workgroups): cycles since the wave's entry   min / median / max over waves" % (T, K, N, len(acc[0])))
    with np.errstate(all="ignore"):
        lo, med, hi = np.nanmin(s, axis=(0, 1)), np.nanmedian(s, axis=(0, 1)), np.nanmax(s, axis=(0, 1))
    for i, nm in enumerate(names):
        print("    %-30s %7.0f %7.0f %7.0f" % (nm, lo[i], med[i], hi[i]))
    # where the spread is: inside a workgroup (its waves finish their K pieces at different times) or between workgroups
    with np.errstate(all="ignore"):
        eol = s[:, :, 7]
        inwg = np.nanmax(eol, axis=1) - np.nanmin(eol, axis=1)
        wgend = np.nanmax(s[:, :, 9], axis=1)
    print("    end of K loop, slowest - fastest wave of a workgroup: median %.0f  p90 %.0f   |  workgroup end (since its entry): p10 %.0f median %.0f p90 %.0f max %.0f"
          % (np.nanmedian(inwg), np.nanpercentile(inwg, 90), np.nanpercentile(wgend, 10), np.nanmedian(wgend), np.nanpercentile(wgend, 90), np.nanmax(wgend)))
    del mods
