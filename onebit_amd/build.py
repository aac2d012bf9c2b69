"""Build libonebit_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is several translation units (one hipcc process each, run in parallel; an object is rebuilt only when its
source or one of the headers it includes -- transitively -- is newer), linked into ONE shared library:
  onebit_hip.hip     the C ABI of rounds 1-5 (layer, decode steps, prefill glue, train-mode layer)
  onebit_mixed.hip   round 6: ragged attention / rope kernels, the split-KV decode attention, the mixed prefill + decode step
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libonebit_hip.so")
SOURCES = ["onebit_hip.hip", "onebit_mixed.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libonebit_hip.so)")


def _deps(path: str, seen=None) -> set:
    """`path` and every file it #includes with quotes, transitively."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale(src: str, extra: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o):
        return True
    stamp = o + ".flags"
    if not os.path.exists(stamp) or open(stamp).read() != extra:
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps(os.path.join(CSRC, src)))


def needs_build() -> bool:
    extra = os.environ.get("OB_EXTRA_HIPCC_FLAGS", "")
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_stale(s, extra) or os.path.getmtime(_obj(s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = False, lib: str = LIB, extra_flags=None, obj_dir: str = OBJ) -> str:
    """Compile the stale translation units (all of them with ``force``) in parallel and link ``lib``.
    ``extra_flags`` / ``obj_dir``: A/B builds with -D switches (tools/variant_bench.py) into their own object directory."""
    global OBJ
    extra = " ".join(extra_flags) if extra_flags is not None else os.environ.get("OB_EXTRA_HIPCC_FLAGS", "")
    keep, OBJ = OBJ, obj_dir
    try:
        os.makedirs(OBJ, exist_ok=True)
        todo = [s for s in SOURCES if force or _stale(s, extra)]
        if not todo and os.path.exists(lib) and all(os.path.getmtime(_obj(s)) <= os.path.getmtime(lib) for s in SOURCES):
            return lib
        cc = _hipcc()

        def compile_one(src):
            cmd = [cc] + FLAGS + extra.split() + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(_obj(src) + ".flags", "w") as f:
                f.write(extra)

        with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
            list(ex.map(compile_one, todo))
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [_obj(s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return lib
    finally:
        OBJ = keep


if __name__ == "__main__":
    import sys
    print(build(force="--incremental" not in sys.argv, verbose=True))
