#!/usr/bin/env python3
"""Print the interesting parts of a bench.py JSON line: python tools/bench_view.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "incomplete", d.get("incomplete"))
r = d.get("roofline") or {}
print("roofline", {k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "traffic", "kernel")})
for k in ("continuous_batch", "decode_ctx"):
    print(k, json.dumps(d.get(k), indent=0))
e = d.get("eval_ppl") or {}
print("eval.ll", json.dumps(e.get("loglikelihood_tokens")))
print("prefill_k_sharded", json.dumps(d.get("prefill_k_sharded"))[:600])
print("prefill_model", json.dumps(d.get("prefill_model"))[:400])
print("decode_k_sharded", json.dumps({k: v for k, v in (d.get("decode_k_sharded") or {}).items() if k in ("fused", "fused_eager", "single_gpu_engine", "ms_per_token", "tokens_per_s")}))
print("errors", {k: v.get("error") for k, v in d.items() if isinstance(v, dict) and "error" in v})
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
