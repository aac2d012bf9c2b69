#!/usr/bin/env python3
"""BASELINE config 5's mixed prefill + decode workload in isolation (bench.py's `continuous_batch.mixed` leg) on a 13B- or 7B-shaped
synthetic checkpoint: python tools/mixed_probe.py [13b|7b] [--step-only].  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel table of the step (profiles/r06_mixed_step_*)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from onebit_amd.llama import build_synthetic_model

name = next((a for a in sys.argv[1:] if not a.startswith("-")), "13b")
dev = torch.device("cuda:0")
model = build_synthetic_model(bench.model_config(name), seed=4242, device=dev)
kw = dict(requests=8, new_tokens=4) if "--step-only" in sys.argv else {}
print(json.dumps(bench.measure_mixed_step(model, dev, **kw), indent=1))
