#!/usr/bin/env python3
"""A/B meter for kernel variants: builds libonebit_hip.so once per set of -D switches (on the build host, where hipcc
cross-compiles) or, on the GPU box, runs bench.py's decode leg + per-launch roofline chain against each prebuilt
variant and prints one line per variant.

  build (no GPU):   python tools/variant_bench.py build name1:-DOB_X=0 name2:"-DOB_X=1 -DOB_Y=2" ...
                    -> onebit_amd/csrc/variants/libonebit_<name>.so   (travels to the GPU box with gpurun)
  run (GPU box):    python tools/variant_bench.py run [--serve] [--prefill] name1 name2 ...
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "onebit_amd", "csrc", "variants")


def build(specs):
    """One library per variant through the package's own builder (all translation units, the variant's -D switches on each,
    objects in a directory of the variant's own)."""
    sys.path.insert(0, ROOT)
    from onebit_amd import build as ob_build
    os.makedirs(VDIR, exist_ok=True)
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(VDIR, "libonebit_%s.so" % name)
        ob_build.build(force=True, lib=out, extra_flags=flags.split(), obj_dir=os.path.join(VDIR, "obj_" + name))
        print("built", name)


def run(args):
    extra = []
    names = []
    for a in args:
        if a == "--serve": extra.append("serve")
        elif a == "--prefill": extra.append("prefill")
        else: names.append(a)
    for name in names:
        lib = os.path.join(VDIR, "libonebit_%s.so" % name) if name != "default" else ""
        env = dict(os.environ)
        if lib:
            env["ONEBIT_LIB"] = lib
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "128", "--warmup", "16", "--no-cpu-baseline", "--no-k-sharded-decode", "--no-eval", "--no-train"]
        if "serve" not in extra: cmd.append("--no-serve")
        if "prefill" not in extra: cmd.append("--no-prefill")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print(name, "FAILED", r.stderr[-600:])
            continue
        per = {k["kernel"]: k["avg_launch_us"] for k in d["roofline"]["per_kernel"]}
        line = "%-14s %8.1f tok/s  %.4f ms  " % (name, d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % kv for kv in per.items())
        if d.get("continuous_batch"): line += "  | serve %.3f ms" % d["continuous_batch"]["ms_per_step"]
        if d.get("prefill_model"): line += "  | prefill_model %.1f ms" % d["prefill_model"]["ms"]
        if d.get("prefill_k_sharded"): line += "  | layer %.0f TF" % d["prefill_k_sharded"].get("token_sharded", {}).get("TFLOPs", 0)
        print(line, flush=True)


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("build", "run"):
        sys.exit(__doc__)
    (build if sys.argv[1] == "build" else run)(sys.argv[2:])
