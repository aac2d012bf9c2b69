#!/usr/bin/env python3
"""Benchmark: LLaMA-7B OneBit greedy decode (BASELINE.json configs[1]) on N MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one decoded token (batch 1) through the whole model: 224 packed 1-bit projections
(the hot path) plus RMSNorm / RoPE / attention over the KV cache / SiLU / residuals / fp16
lm_head / argmax.  Weights are a synthetic 7B-shaped OneBit inference checkpoint in the
reference's layout, resident in HBM before timing.  For N > 1 every rank decodes its own
sequence on its own full replica (decode does not shard, BASELINE.json north_star: "decode
stays single-GPU") -- weak scaling, no data-path collective; `value` is the aggregate.

With --gpus N > 1 and no WORLD_SIZE in the environment the script re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU over RCCL); launched by
torchrun directly it uses the ranks it is given.  Either way it refuses to report n_gpus != the
RCCL world size.

The JSON line also carries
  roofline     -- the TIME-dominant kernel of the decode step (one of the four fused 1-bit GEMV
                  launches of a layer), each timed with HIP events over the 32 layers' own weights
                  (> 256 MB Infinity Cache), algorithmic bytes / time against the 8 TB/s HBM peak;
                  `per_kernel` lists all four, `whole_token` the fraction of the complete step
  cpu_baseline -- the oracle's reference-style CPU path (unpack every call) timed on this host: one
                  core (C port), all cores (the reference's own ATen op sequence) and the same with
                  the dense matrix unpacked once
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def csrc_sha():
    """Hash of the kernel sources: stamps PMC-derived numbers so a stale file is never reported."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "onebit_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def spawn_command(gpus, argv, port=None):
    """The torchrun command `bench.py --gpus N` re-executes itself under when it was started as a
    plain process (one rank per GPU, rendezvous on 127.0.0.1)."""
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--engine", default="auto", choices=["auto", "eager", "fused"])
    ap.add_argument("--prompt", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-prefill", action="store_true")
    ap.add_argument("--no-serve", action="store_true", help="skip the continuous-batching (config 5) field")
    ap.add_argument("--no-train", action="store_true", help="skip the train-mode layer field (forward + backward of BitLinear at a 7B MLP shape)")
    ap.add_argument("--no-eval", action="store_true", help="skip the evaluation-caller field (perplexity windows + ragged loglikelihood batch)")
    ap.add_argument("--no-k-sharded-decode", action="store_true",
                    help="skip BASELINE config 4 (LLaMA-13B shapes, module-path decode with every 1-bit layer K-sharded "
                         "over the ranks, one all-reduce per BitLinearInf call; extra JSON field, not the headline value)")
    ap.add_argument("--deadline", type=float, default=1500.0,
                    help="seconds after the headline measurement at which rank 0 prints the JSON line with whatever legs have finished "
                         "(field `incomplete`) and every rank exits: a hung secondary leg (a collective that never returns) cannot suppress the line")
    ap.add_argument("--pg-timeout", type=float, default=300.0, help="process-group timeout in seconds (N > 1)")
    return ap.parse_args(argv)


def algorithmic_bytes(T, K, N):
    """packed W + h + g + T * (x + y), fp16 I/O."""
    return N * K // 8 + 2 * K + 2 * N + T * (2 * K + 2 * N)


def model_config(name):
    from onebit_amd.llama import OneBitLlamaConfig
    if name == "7b":
        return OneBitLlamaConfig.llama_7b()
    if name == "13b":
        return OneBitLlamaConfig.llama_13b()
    return OneBitLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=4,
                             num_attention_heads=8, max_position_embeddings=256)


def token_bytes(cfg, ctx):
    """Algorithmic HBM bytes of one decoded token: 1-bit layers + lm_head + KV read."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    per_layer = 4 * algorithmic_bytes(1, H, H) + 2 * algorithmic_bytes(1, H, I) + algorithmic_bytes(1, I, H)
    star = per_layer * cfg.num_hidden_layers
    lm_head = 2 * cfg.vocab_size * H
    kv = 2 * 2 * H * ctx * cfg.num_hidden_layers
    return star, star + lm_head + kv


def _chain_us(launch_layer, layers, dev, min_launches=512):
    """us per launch of `launch_layer(layer)` captured once per decoder layer (distinct weights) in a
    HIP graph and replayed; HIP events on the replay stream.  Includes the dependent-kernel boundary
    every launch of a decode chain pays."""
    us, n = _chain_total_us([launch_layer], layers, dev, min_launches)
    return us / len(layers), n


def _chain_total_us(fns, layers, dev, min_launches=512, rounds=5):
    """us per REPLAY of the chain `for layer: for fn in fns: fn(layer)` (one HIP graph), median of `rounds` timed
    batches of replays, and the number of launches timed per batch."""
    for layer in layers:
        for fn in fns:
            fn(layer)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for layer in layers:
            for fn in fns:
                fn(layer)
    for _ in range(8):                      # settle clocks / TLBs on the new buffers before timing
        graph.replay()
    torch.cuda.synchronize(dev)
    nl = len(layers) * len(fns)
    reps = max(2, min_launches // nl)
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts), reps * nl


def measure_roofline(model, dev, ms_per_token, tok_bytes):
    """The four fused 1-bit GEMV launches of a decoder layer, exactly as onebit_decode_step issues
    them (prologues, producers' tile partials), each as a chain over the layers' own weights.  The
    `roofline` object is the one that takes the most time per token; `per_kernel` lists all four;
    attention + lm_head + argmax are the remainder of the measured step."""
    from onebit_amd.engine import PRO_PLAIN, PRO_RES_LN_RMS, PRO_SWIGLU, fused_gemv, tile_stats_floats
    cfg = model.config
    H, I = cfg.hidden_size, cfg.intermediate_size
    f16 = torch.float16
    rn = lambda n: torch.randn(n, device=dev).to(f16)
    hres, u_prev, attn = rn(H), rn(H), rn(H)
    ug_in, uu_in = rn(I), rn(I)
    hout = torch.empty(H, device=dev, dtype=f16)
    o = {k: torch.empty(n, device=dev, dtype=f16) for k, n in dict(q=H, k=H, v=H, o=H, g=I, u=I, d=H).items()}
    st = {k: torch.zeros(tile_stats_floats(n), device=dev) for k, n in dict(h=H, q=H, k=H, v=H, o=H, g=I, u=I, d=H, gi=I, ui=I).items()}
    layers = list(model.model.layers)
    L = len(layers)
    if os.environ.get("OB_BENCH_REUSE_WEIGHTS"):        # diagnostic: every launch of a chain on ONE layer's matrices (L2-resident rows)
        layers = [layers[0]] * L
    ab = algorithmic_bytes
    kinds = {
        "qkv": (lambda l: fused_gemv([l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj], [o["q"], o["k"], o["v"]],
                                     PRO_RES_LN_RMS, rms_eps=cfg.rms_norm_eps, hres_in=hres, u_prev=u_prev, hres_out=hout,
                                     rms_w=l.input_layernorm.weight, st_prev=st["h"], stats_out=[st["q"], st["k"], st["v"]]),
                3 * ab(1, H, H), "residual + LayerNorm + RMSNorm prologue, q|k|v, %d->3x%d" % (H, H)),
        "o": (lambda l: fused_gemv([l.self_attn.o_proj], [o["o"]], PRO_PLAIN, xin=attn, stats_out=[st["o"]]),
              ab(1, H, H), "o_proj %d->%d" % (H, H)),
        "gate_up": (lambda l: fused_gemv([l.mlp.gate_proj, l.mlp.up_proj], [o["g"], o["u"]], PRO_RES_LN_RMS,
                                         rms_eps=cfg.rms_norm_eps, hres_in=hres, u_prev=u_prev, hres_out=hout,
                                         rms_w=l.post_attention_layernorm.weight, st_prev=st["h"], stats_out=[st["g"], st["u"]]),
                    2 * ab(1, H, I), "residual + LayerNorm + RMSNorm prologue, gate|up, %d->2x%d" % (H, I)),
        "down": (lambda l: fused_gemv([l.mlp.down_proj], [o["d"]], PRO_SWIGLU, u_gate=ug_in, u_up=uu_in,
                                      st_gate=st["gi"], st_up=st["ui"], stats_out=[st["d"]]),
                 ab(1, I, H), "SiLU(LayerNorm(gate)) * LayerNorm(up) prologue, down %d->%d" % (I, H)),
    }
    # tile partials the prologues read: produce them once from the synthetic inputs
    fused_gemv([layers[0].self_attn.o_proj], [o["o"]], PRO_PLAIN, xin=attn, stats_out=[st["h"]])
    u_prev.copy_(o["o"])
    fused_gemv([layers[0].mlp.gate_proj, layers[0].mlp.up_proj], [o["g"], o["u"]], PRO_RES_LN_RMS, rms_eps=cfg.rms_norm_eps,
               hres_in=hres, u_prev=u_prev, hres_out=hout, rms_w=layers[0].post_attention_layernorm.weight, st_prev=st["h"],
               stats_out=[st["gi"], st["ui"]])
    ug_in.copy_(o["g"]); uu_in.copy_(o["u"])
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("csrc_sha") == csrc_sha():
                traffic = tj.get("bytes_per_launch", {})
        except Exception:
            traffic = {}
    # Cache honesty (>= 360 MB of distinct weights per chain): the four launches are timed IN LAYER ORDER
    # over the whole model -- one graph of qkv, o, gate_up, down for each of the L layers (810 MB of packed rows at 7B,
    # 3x the Infinity Cache; a chain of one kind alone cycles through 68-361 MB and may be served from it) -- and a
    # launch's time is what the chain loses when that launch is left out: (T_all - T_without_k) / L.
    order = ["qkv", "o", "gate_up", "down"]
    t_all, n_all = _chain_total_us([kinds[k][0] for k in order], layers, dev)
    per = []
    for name in order:
        fn, nbytes, what = kinds[name]
        t_wo, _ = _chain_total_us([kinds[k][0] for k in order if k != name], layers, dev)
        us = max((t_all - t_wo) / L, 1e-3)
        us_iso, n_iso = _chain_us(fn, layers, dev)
        gbs = nbytes / (us * 1e-6) / 1e9
        per.append({"kernel": name, "what": what, "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(us, 3),
                    "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "us_per_token": round(us * L, 1),
                    "traffic": traffic.get(name), "launches": n_all // len(order),
                    "isolated_chain_us": round(us_iso, 3)})
    dom = max(per, key=lambda r: r["us_per_token"])
    gemv_us = sum(r["us_per_token"] for r in per)
    whole = tok_bytes / (ms_per_token * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "ob_dec_gemv_kernel: %s (%s), T=1" % (dom["kernel"], dom["what"]),
            "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": dom["traffic"],
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "avg_launch_us": dom["avg_launch_us"],
            "launches": dom["launches"], "distinct_weight_sets": L, "selection": "largest time per token of the four GEMV launches",
            "timing": "HIP events over graph replays of the four launches in LAYER ORDER over all %d layers' weights (%.0f MB of "
                      "packed rows per replay); a launch's time = (chain - chain without it) / layers; isolated_chain_us = "
                      "the same launch as a chain of its own kind (rounds 1-3 figure)" % (L, sum(k[1] for k in kinds.values()) * L / 1e6),
            "layer_chain_us": round(t_all / L, 3),
            "per_kernel": per,
            "whole_token": {"algorithmic_bytes": tok_bytes, "ms": round(ms_per_token, 4), "achieved": round(whole, 1),
                            "frac": round(whole / HBM_PEAK_GBS, 4),
                            "gemv_launches_us": round(gemv_us, 1),
                            "attention_lm_head_argmax_us": round(ms_per_token * 1e3 - gemv_us, 1)},
            "traffic_source": ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, same kernel sources)" if traffic
                               else "none for these kernel sources (profiles/pmc_traffic.json is stamped with another csrc_sha)"),
            "note": "HIP events around graph replays of back-to-back launches: includes the ~1.7 us kernel boundary"}


def _timed(fn, dev, world, warm=6, iters=20):
    """Per-iteration HIP-event times of fn() (>= 20 iterations after `warm` warm-ups): returns
    (median_s, min_s) of the slowest rank's figures."""
    import torch.distributed as dist
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e-3 for i in range(iters)]
    med, mn = statistics.median(ts), min(ts)
    if world > 1:
        t = torch.tensor([med, mn], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med, mn = float(t[0]), float(t[1])
    return med, mn


def measure_prefill_sharded(cfg, dev, world, rank):
    """K-sharded (hidden-dim sharded) prefill of one 1-bit layer, hidden -> intermediate, T = 8 x 2048
    tokens (BASELINE configs[2] shape): every rank holds K/world columns of the packed matrix,
    computes fp32 partial sums on its MFMA kernel, reduce-scatters them over tokens (RCCL),
    applies g + LayerNorm to its rows and all-gathers the fp16 result (onebit_amd/sharded.py)."""
    import torch.distributed as dist
    from onebit_amd.sharded import k_sharded_forward, shard_k
    K, N, T = cfg.hidden_size, cfg.intermediate_size, 8 * 2048
    g = torch.Generator(device=dev).manual_seed(77)
    W = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8, device=dev).view(torch.int8)
    h = (0.1 * (0.5 + torch.rand(K, generator=g, device=dev))).half()
    gs = (0.1 * (0.5 + torch.rand(N, generator=g, device=dev))).half()
    x = torch.randn(T, K, generator=g, device=dev).half()
    shard = shard_k(W, h, gs, None, rank, world)
    flops = 2.0 * T * K * N

    def entry(dt, dmin, **kw):
        d = {"ms_per_call": round(dt * 1e3, 3), "ms_min": round(dmin * 1e3, 3), "iterations": 20,
             "tokens_per_s": round(T / dt, 1), "TFLOPs": round(flops / dt / 1e12, 1), "TFLOPs_best": round(flops / dmin / 1e12, 1),
             "frac_of_mfma_peak": round(flops / dt / 1e12 / (2500.0 * world), 4)}
        d.update(kw)
        return d

    dt, dmin = _timed(lambda: k_sharded_forward(shard, x, mode="rs_ag"), dev, world)
    res = entry(dt, dmin, layer="%d->%d" % (K, N), tokens=T, k_shards=world, exchange="reduce_scatter(fp32)+all_gather(fp16)",
                mfma_peak_TFLOPs=2500.0 * world, timing="median of 20 HIP-event-timed calls (slowest rank)")
    # the layout that needs no exchange at all: the packed matrix is only N*K/8 bytes, so every rank
    # keeps all of it and takes T/world tokens (token sharding); reported beside the K-sharded path
    from onebit_amd import BitLinearInf
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data, m.input_factor.data, m.weight_scale.data = W, h, gs
    Tl = T // world
    xl = x[rank * Tl:(rank + 1) * Tl]
    dt2, dmin2 = _timed(lambda: m(xl), dev, world)
    # N-sharded (output rows split): only the LayerNorm statistics cross ranks (one all-gather of [T, 2] fp32)
    try:
        from onebit_amd.sharded import n_sharded_forward, shard_n
        nsh = shard_n(W, h, gs, None, rank, world)
        dt3, dmin3 = _timed(lambda: n_sharded_forward(nsh, x), dev, world, warm=4)
        res["n_sharded"] = entry(dt3, dmin3, rows_per_rank=nsh.n1 - nsh.n0,
                                 exchange="all_gather(fp32 [T,2] row statistics); output stays N-sharded")
    except Exception as e:
        res["n_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}
    res["token_sharded"] = entry(dt2, dmin2, tokens_per_rank=Tl)
    # context, not a baseline the metric names: the vendor library's DENSE fp16 GEMM of the same shape on already-unpacked weights (what the
    # reference's forward costs after its unpack), timed the same way in the same process
    try:
        wd = torch.randn(N, K, generator=g, device=dev).half()
        yd = torch.empty(Tl, N, device=dev, dtype=torch.float16)
        dt4, dmin4 = _timed(lambda: torch.matmul(xl, wd.t(), out=yd), dev, world)
        res["vendor_dense_fp16_gemm"] = entry(dt4, dmin4, tokens_per_rank=Tl, what="torch.matmul (hipBLASLt / rocBLAS) on dense fp16 weights, no scales, no LayerNorm",
                                              sustained="profiles/r06_dense_vs_onebit_power.txt: 1273 TFLOP/s at 1.84 GHz against 1481 for the 1-bit GEMM alone, both at the 1.4 kW limit")
        del wd, yd
    except Exception as e:
        res["vendor_dense_fp16_gemm"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def measure_k_sharded_decode(cfg, dev, world, rank, steps, prompt_len):
    """BASELINE config 4 (LLaMA-13B shapes): greedy decode with every BitLinearInf K-sharded (onebit_amd/sharded.py,
    KShardedBitLinear): partial GEMV on the rank's K slice, all-reduce of the [1, N] fp32 partials, g + LayerNorm
    everywhere.  Three figures:
      single_gpu_engine  the fused DecodeEngine on the WHOLE 13B checkpoint on one GPU -- the honest N = 1 line any
                         sharded figure has to be compared with (every rank runs it on its own copy);
      graph              the sharded step under ONE HIP graph with the all-reduces captured (StaticShapeDecoder:
                         device-side token / position, static shapes) -- measures the exchange, not Python;
      eager              the same sharded modules driven from Python (the exchange test-bed of rounds 1-3; ~600
                         host-issued launches per token: host-bound, NOT a baseline)."""
    import torch.distributed as dist
    from onebit_amd.llama import build_synthetic_model
    from onebit_amd.sharded import StaticShapeDecoder, shard_model_k
    model = build_synthetic_model(cfg, seed=4242, device=dev)          # same checkpoint on every rank
    g = torch.Generator(device="cpu").manual_seed(7)
    prompt = torch.randint(0, cfg.vocab_size, (1, prompt_len), generator=g).to(dev)
    max_len = prompt_len + 2 * steps + 16

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(step_fn, warm):
        for _ in range(warm):
            step_fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    out = {"k_shards": world, "exchange": "fused: all_reduce(fp32) of z_qkv / z_o / z_gu / z_down per layer (4); module paths: one per BitLinearInf call (7)", "steps": steps,
           "collectives_per_token": 7 * cfg.num_hidden_layers if world > 1 else 0}
    try:                                                               # the fused engine on the unsharded checkpoint
        from onebit_amd.engine import DecodeEngine
        eng = DecodeEngine(model, max_len=max_len)
        eng.prefill(prompt)
        dt = timed(eng.step, 4)
        out["single_gpu_engine"] = {"ms_per_token": round(dt / steps * 1e3, 4), "tokens_per_s": round(steps / dt, 1),
                                    "engine": "onebit_decode_step (HIP graph), whole checkpoint on one GPU",
                                    "note": "the N = 1 line sharded decode compares with"}
        del eng
    except Exception as e:
        out["single_gpu_engine"] = {"error": "%s: %s" % (type(e).__name__, e)}
    def agree(ok):
        """True only if `ok` on EVERY rank: a rank that failed a phase must not leave the others alone in the next collective."""
        if world <= 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    # round 5: native segments (onebit_decode_step_ksharded), q|k|v and gate|up one exchange each.  Round 6: the EAGER (uncaptured)
    # run first -- its number stands if capturing the collectives into a HIP graph fails on any rank (agreed by an all-reduce of a
    # flag before anybody replays)
    fdec, err = None, None
    try:
        from onebit_amd.sharded import FusedKShardedDecoder
        fdec = FusedKShardedDecoder(model, rank, world, max_len=max_len, use_graph=False)
        fdec.prime(prompt)                                         # (replicated module path: no collective inside)
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    if not agree(err is None):
        out["fused"] = {"error": err or "another rank failed to build the fused decoder"}
    else:
        try:
            L = cfg.num_hidden_layers
            dt = timed(fdec.step, 3)
            fused = {"ms_per_token": round(dt / steps * 1e3, 4), "tokens_per_s": round(steps / dt, 1),
                     "path": "FusedKShardedDecoder, eager: onebit_decode_step_ksharded segments + all_reduce(fp32), host-issued",
                     "collectives_per_token": fdec.collectives_per_token, "launches_per_token": 8 * L + 3,
                     "reduced_fp32_bytes_per_token": 4 * L * (cfg.num_attention_heads * cfg.head_dim + 2 * cfg.num_key_value_heads * cfg.head_dim
                                                              + 2 * cfg.hidden_size + 2 * cfg.intermediate_size) if world > 1 else 0}
            out["fused_eager"] = dict(fused)
            fdec.use_graph = True
            if agree(fdec.capture()):
                dt = timed(fdec.step, 3)
                fused.update(ms_per_token=round(dt / steps * 1e3, 4), tokens_per_s=round(steps / dt, 1),
                             path="FusedKShardedDecoder: onebit_decode_step_ksharded segments (decode GEMV in fp32-partial form on the rank's "
                                  "K slice, row kernels, decode attention) + all_reduce(fp32) replayed as one HIP graph")
            else:
                fdec.use_graph, fdec.graph = False, None
                fused["graph_capture"] = "failed on at least one rank: the eager figure stands"
            out["fused"] = fused
        except Exception as e:
            out["fused"] = {"error": "%s: %s" % (type(e).__name__, e)}
    del fdec
    torch.cuda.empty_cache()
    shard_model_k(model, rank, world, mode="allreduce")
    torch.cuda.empty_cache()
    try:                                                               # sharded step as one HIP graph, collectives captured
        dec = StaticShapeDecoder(model, max_len=max_len, use_graph=True)
        dec.prime(prompt)
        dt = timed(dec.step, 3)
        out["graph"] = {"ms_per_token": round(dt / steps * 1e3, 3), "tokens_per_s": round(steps / dt, 2),
                        "path": "StaticShapeDecoder: module kernels + RCCL all-reduces replayed as one HIP graph"}
        del dec
    except Exception as e:
        out["graph"] = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.cuda.empty_cache()
    cache = model.new_cache(1, max_len)
    state = {"tok": model(prompt, cache)[:, -1].argmax(-1, keepdim=True)}

    def eager_step():
        state["tok"] = model(state["tok"], cache)[:, -1].argmax(-1, keepdim=True)
    dt = timed(eager_step, 2)
    out["eager"] = {"ms_per_token": round(dt / steps * 1e3, 3), "tokens_per_s": round(steps / dt, 2),
                    "path": "module path driven from Python (host-bound exchange test-bed, not a baseline)"}
    # headline fields of this object: the fused figure when it exists, else the module graph, else eager
    best = out["fused"] if "error" not in out["fused"] else (out["graph"] if "error" not in out["graph"] else out["eager"])
    out["collectives_per_token"] = best.get("collectives_per_token", out["collectives_per_token"])
    out["ms_per_token"], out["tokens_per_s"], out["path"] = best["ms_per_token"], best["tokens_per_s"], best["path"]
    return out


def measure_continuous_batch(model, dev, slots=32, steps=40):
    """BASELINE config 5 shape on this model: `slots` sequences decoding together through
    onebit_decode_step_batched (one HIP-graph replay per step); steady-state decode steps only, driven the way
    `ContinuousBatcher.run` drives them (bursts of steps between host synchronisations)."""
    from onebit_amd.serving import ContinuousBatcher
    cfg = model.config
    g = torch.Generator(device="cpu").manual_seed(3)
    budget = steps + 64
    cb = ContinuousBatcher(model, max_batch=slots, max_len=16 + budget + 16)
    for _ in range(slots):
        cb.add_request(torch.randint(0, cfg.vocab_size, (16,), generator=g).tolist(), budget)
    while cb.steps < 20:
        cb.step()                                   # prefill step, graph captures, warm replays
    torch.cuda.synchronize(dev)
    s0 = cb.steps
    t0 = time.perf_counter()
    while cb.steps - s0 < steps:
        cb.step()
    torch.cuda.synchronize(dev)
    n = cb.steps - s0
    dt = (time.perf_counter() - t0) / n
    return {"slots": slots, "ms_per_step": round(dt * 1e3, 3), "tokens_per_s": round(slots / dt, 1), "steps_timed": n,
            "engine": "onebit_decode_step_batched" if cb._native is not None else "torch glue",
            "host_syncs": "one per burst of up to %d steps (tokens fed back on the device)" % cb.max_burst,
            "note": "steady-state decode, all slots active, greedy; per GPU"}


def measure_mixed_step(model, dev, slots=32, n_prefill=8, prompt=512, ctx=128, iters=8, requests=128, new_tokens=64):
    """BASELINE config 5's defining workload ("32 sequences with mixed lengths, e.g. 8 prefill of 512 + 24 decode")
    through onebit_mixed_step: (a) ONE step of `n_prefill` x `prompt` prompt tokens next to `slots - n_prefill` decoding requests
    that each hold `ctx` cached tokens, HIP-event timed; (b) a closed-loop request stream (`requests` prompts of 64..`prompt`
    tokens, `new_tokens` greedy tokens each, `slots` KV-cache slots, chunked prefill) through ContinuousBatcher: end-to-end
    generated tokens/s and where the time went (steps with prompt tokens vs decode-only graph steps)."""
    from onebit_amd.engine import MixedStep, fp16_view
    from onebit_amd.serving import ContinuousBatcher
    cfg = model.config
    g = torch.Generator(device="cpu").manual_seed(11)
    V = cfg.vocab_size
    max_len = prompt + new_tokens + 64
    n_dec = slots - n_prefill
    caches = fp16_view(model).new_cache(slots, max_len).layers
    ms = MixedStep(model, caches, slots, max_len, max_rows=n_prefill * prompt + n_dec)
    rnd = lambda n: torch.randint(0, V, (n,), generator=g).tolist()
    ms.launch([(s, 0, rnd(ctx)) for s in range(n_dec)])                       # the decoding requests' history
    items = [(s, ctx, rnd(1)) for s in range(n_dec)] + [(n_dec + i, 0, rnd(prompt)) for i in range(n_prefill)]
    rows = n_dec + n_prefill * prompt
    for _ in range(2):
        ms.launch(items)                                                       # (the same cache rows are rewritten: idempotent)
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        ms.launch(items)
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    med = ts[len(ts) // 2]
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    w1 = L * (4 * H * H + 3 * H * I)
    step = {"decode_rows": n_dec, "decode_context": ctx, "prefill_segments": n_prefill, "prefill_tokens_each": prompt, "rows": rows,
            "ms": round(med, 3), "ms_min": round(ts[0], 3), "iterations": iters, "tokens_per_s": round(rows / med * 1e3, 1),
            "onebit_layer_TFLOPs_equivalent": round(2.0 * rows * w1 / med / 1e9, 1),
            "launches_per_layer": "norm, q|k|v (one grouped GEMM), rope / append, flash (prompt chunks), key-block (decode rows), o, norm, gate|up (grouped), swiglu, down = 10",
            "engine": "onebit_mixed_step"}
    # (a2) chunked prefill with a small step budget: ONE prompt chunk of 64 / 256 / 512 tokens next to slots - 1 decoding requests -- the
    # steps a latency-minded scheduler issues (routes: passes of the skinny GEMM / grouped GEMM + K-sliced o, down / + skinny tail)
    mid = []
    if requests >= 32:                             # (the probes under tools/ that profile ONE step shape pass a short stream)
        nd = slots - 1
        ms.launch([(s, 0, rnd(ctx)) for s in range(nd)])
        for chunk in (64, 256, 512):
            if chunk > prompt:
                continue
            it2 = [(s, ctx, rnd(1)) for s in range(nd)] + [(nd, 0, rnd(chunk))]
            for _ in range(2):
                ms.launch(it2)
            torch.cuda.synchronize(dev)
            e2 = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            e2[0].record()
            for i in range(4):
                ms.launch(it2)
                e2[i + 1].record()
            torch.cuda.synchronize(dev)
            t2 = sorted(e2[i].elapsed_time(e2[i + 1]) for i in range(4))
            mid.append({"rows": nd + chunk, "decode_rows": nd, "prompt_chunk": chunk, "ms": round(t2[1], 3), "tokens_per_s": round((nd + chunk) / t2[1] * 1e3, 1)})
    step["mid_size_steps"] = mid
    del ms, caches
    torch.cuda.empty_cache()
    # (b) the request stream
    lens = torch.randint(64, prompt + 1, (requests,), generator=g).tolist()
    cb = ContinuousBatcher(model, max_batch=slots, max_len=max_len, prefill_chunk=prompt, max_step_tokens=n_prefill * prompt + slots)
    for n in lens[:4]:                                                         # warm-up: graph capture, lazy initialisation
        cb.add_request(rnd(n), 4)
    cb.run()
    cb.steps = cb.mixed_steps = cb.graph_steps = cb.tokens_scheduled = 0
    cb.time_mixed = cb.time_decode = 0.0
    for n in lens:
        cb.add_request(rnd(n), new_tokens)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = cb.run()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gen = sum(len(v) for v in out.values()) - 4 * 4
    reqs = [r for r in cb.sched.finished.values() if r.max_new_tokens == new_tokens]
    ttft = sorted(r.t_first - r.t_add for r in reqs)
    itl = sorted((r.t_done - r.t_first) / max(len(r.out) - 1, 1) for r in reqs)
    pct = lambda v, q: round(v[min(len(v) - 1, int(q * len(v)))] * 1e3, 2)
    stream = {"requests": requests, "prompt_tokens": "64..%d (mean %.0f)" % (prompt, sum(lens) / len(lens)), "new_tokens_each": new_tokens,
              "slots": slots, "prefill_chunk": prompt, "max_step_tokens": n_prefill * prompt + slots, "wall_s": round(dt, 3),
              "generated_tokens_per_s": round(gen / dt, 1), "all_tokens_per_s": round((gen + sum(lens)) / dt, 1),
              "steps": cb.steps, "steps_with_prompt_tokens": cb.mixed_steps, "decode_only_graph_steps": cb.graph_steps,
              "time_share_mixed_steps": round(cb.time_mixed / max(cb.time_mixed + cb.time_decode, 1e-9), 3),
              "time_share_decode_steps": round(cb.time_decode / max(cb.time_mixed + cb.time_decode, 1e-9), 3),
              "ms_per_mixed_step": round(cb.time_mixed / max(cb.mixed_steps, 1) * 1e3, 2),
              "ms_per_decode_step": round(cb.time_decode / max(cb.graph_steps, 1) * 1e3, 3),
              "time_to_first_token_ms": {"p50": pct(ttft, 0.5), "p99": pct(ttft, 0.99), "note": "all requests submitted at t = 0 (closed loop): includes the queueing for a slot"},
              "inter_token_latency_ms": {"p50": pct(itl, 0.5), "p99": pct(itl, 0.99)},
              "engine": "onebit_mixed_step + onebit_decode_step_batched (HIP graph)" if cb._mixed is not None and cb._native is not None else "torch glue"}
    return {"step": step, "request_stream": stream, "per": "GPU", "data": "synthetic"}


def measure_decode_ctx(model, dev, contexts=(512, 2000), slots=32, slot_ctx=512, steps=32):
    """Decode against LONG contexts (the headline runs at <= 42 cached tokens): single stream through DecodeEngine at `contexts`
    cached tokens (key-block attention graphs: onebit_rows_qkv_rope_ragged + onebit_attention_decode_rows), and the `slots`-slot
    batched step at `slot_ctx` cached tokens per slot in both attention forms.  KV rate = the K / V bytes of the context over the
    time the step takes BEYOND the same step at 16 cached tokens (what streaming the context costs)."""
    from onebit_amd.engine import BatchedDecodeStep, DecodeEngine
    cfg = model.config
    H, D, L = cfg.num_attention_heads, cfg.head_dim, cfg.num_hidden_layers
    max_len = min(cfg.max_position_embeddings, max(max(contexts), slot_ctx) + steps + 16)
    eng = DecodeEngine(model, max_len=max_len)
    for kc, vc in eng.cache.layers:
        kc.normal_(); vc.normal_()

    def single(ctx):
        eng.set_state(5, ctx)
        for _ in range(4):
            eng.step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps
    base = single(16)
    one = {"cached_tokens_16": {"ms_per_token": round(base * 1e3, 4)}}
    for ctx in contexts:
        ctx = min(ctx, max_len - steps - 8)
        t = single(ctx)
        kv = L * cfg.num_key_value_heads * (ctx + steps // 2) * D * 2 * 2
        one["cached_tokens_%d" % ctx] = {"ms_per_token": round(t * 1e3, 4), "tokens_per_s": round(1.0 / t, 1), "kv_bytes_per_token": kv,
                                        "kv_GBps_over_the_short_step": round(kv / max(t - base, 1e-9) / 1e9, 1),
                                        "frac_of_8TBps": round(kv / max(t - base, 1e-9) / 1e9 / HBM_PEAK_GBS, 3)}
    one["attention"] = "key-block: rope / append launch + onebit_attention_decode_rows (%d positions per workgroup), one HIP graph per split count" % eng._kb_chunk \
        if eng._keyblock else "scores + P.V kernel pair"
    del eng
    torch.cuda.empty_cache()
    caches = model.new_cache(slots, max_len).layers
    for kc, vc in caches:
        kc.normal_(); vc.normal_()

    def batched(step, ctx):
        step.tokens.fill_(5)
        step.pos.fill_(ctx - 1)
        step.launch()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step.launch()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            g.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / steps
    ns = max(1, -(-slot_ctx // 512))
    short = BatchedDecodeStep(model, caches, slots, max_len)
    kb = BatchedDecodeStep(model, caches, slots, max_len, attn_splits=ns, attn_chunk=512)
    b16 = batched(short, 16)
    t_short, t_kb = batched(short, slot_ctx), batched(kb, slot_ctx)
    kvb = slots * L * cfg.num_key_value_heads * slot_ctx * D * 2 * 2
    bat = {"slots": slots, "cached_tokens_per_slot": slot_ctx, "kv_bytes_per_step": kvb, "ms_per_step_at_16_cached": round(b16, 3),
           "one_workgroup_per_head_slot": {"ms_per_step": round(t_short, 3), "kv_GBps_over_the_short_step": round(kvb / max(t_short - b16, 1e-9) / 1e6, 1)},
           "key_block": {"ms_per_step": round(t_kb, 3), "splits": ns, "chunk": 512,
                         "kv_GBps_over_the_short_step": round(kvb / max(t_kb - b16, 1e-9) / 1e6, 1),
                         "frac_of_8TBps": round(kvb / max(t_kb - b16, 1e-9) / 1e6 / HBM_PEAK_GBS, 3)},
           "tokens_per_s": round(slots / min(t_short, t_kb) * 1e3, 1)}
    return {"single_stream": one, "batched": bat, "per": "GPU", "data": "synthetic (random K / V rows)"}


def measure_ttft(model, dev, prompts=(16, 128, 512, 2000)):
    """Time to first token of ONE request: DecodeEngine.prime (the prompt through onebit_mixed_step: one C call, lm_head on the last row)
    beside DecodeEngine.prefill (module path with the fused glue, logits of every row); wall clock around call + device sync, median of 5."""
    from onebit_amd.engine import DecodeEngine
    cfg = model.config
    max_len = min(cfg.max_position_embeddings, max(prompts) + 48)
    eng = DecodeEngine(model, max_len=max_len)
    g = torch.Generator().manual_seed(3)

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t0)
        return round(sorted(ts)[2] * 1e3, 3)

    res = {"engine": "DecodeEngine.prime = onebit_mixed_step over the prompt rows", "per_prompt": []}
    for S in prompts:
        S = min(S, max_len - 8)
        ids = torch.randint(0, cfg.vocab_size, (1, S), generator=g).to(dev)
        native = timed(lambda: eng.prime(ids))
        first = eng.first_token
        module = timed(lambda: eng.prefill(ids))
        res["per_prompt"].append({"prompt_tokens": S, "ms": native, "prompt_tokens_per_s": round(S / native * 1e3, 1), "module_path_ms": module,
                                  "first_token_equal": bool(first == eng.first_token)})
    del eng
    return res


def measure_prefill_model(model, dev, B=8, S=2048):
    """BASELINE configs[2]: whole-model prefill of B x S tokens (1-bit GEMMs + fused row glue +
    the vendor's fused attention), tokens/s and the 1-bit layers' share expressed in TFLOP/s."""
    cfg = model.config
    S = min(S, cfg.max_position_embeddings)
    ids = torch.randint(0, cfg.vocab_size, (B, S), generator=torch.Generator(device="cpu").manual_seed(5)).to(dev)
    res = {}
    for impl in ("hip", "sdpa"):
        model.set_attention(impl).set_fused_glue(True)
        try:
            with torch.no_grad():
                model(ids[:1, :128])
                res[impl] = _timed(lambda: model(ids), dev, 1, warm=2, iters=20 if impl == "hip" else 6)
        finally:
            model.set_attention("eager").set_fused_glue(False)
            torch.cuda.empty_cache()
    dt, dmin = res["hip"]
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    w1 = L * (4 * H * H + 3 * H * I)
    return {"batch": B, "seq_len": S, "ms": round(dt * 1e3, 1), "ms_min": round(dmin * 1e3, 1), "iterations": 20,
            "tokens_per_s": round(B * S / dt, 1),
            "onebit_layer_TFLOPs_equivalent": round(2.0 * B * S * w1 / dt / 1e12, 1),
            "attention": "onebit_attention_prefill (own causal flash kernel on MFMA, writes o_proj's pre-scaled rows)",
            "ms_with_torch_sdpa_attention": round(res["sdpa"][0] * 1e3, 1),
            "glue": "onebit_rows_res_ln_rms (writes the consumers' pre-scaled rows) + onebit_rows_qkv_rope + "
                    "onebit_rows_swiglu; projections with ONEBIT_FLAG_PRESCALED", "per": "GPU"}


def measure_prefill_model_tp(model, dev, world, rank, B=8, S=2048, tp_kwargs=None, timed=None):
    """BASELINE configs[2] under model-level tensor parallelism (onebit_amd/tp.py): q|k|v by head ->
    local attention -> o K-sharded, gate|up N-sharded -> down K-sharded; two activation exchanges
    (reduce_scatter fp32 + all_gather fp16 of [T, hidden]) per layer over RCCL.  Every rank holds the
    same checkpoint here (seeded identically by the caller) and keeps 1/world of the 1-bit weights."""
    from onebit_amd.tp import TensorParallelPrefill
    cfg = model.config
    S = min(S, cfg.max_position_embeddings)
    ids = torch.randint(0, cfg.vocab_size, (B, S), generator=torch.Generator(device="cpu").manual_seed(5)).to(dev)
    # (tp_kwargs / timed: the world-2 gloo test runs THIS function on CPU with stand-in compute callbacks and a wall-clock
    # timer, so the control flow the driver launches on N GPUs has executed with more than one rank)
    tp = TensorParallelPrefill(model, rank, world, **(tp_kwargs or dict(attention="hip")))
    timed = timed or _timed
    try:
        tp(ids[:1, :min(128, S)], gather_logits=False)
        dt, dmin = timed(lambda: tp(ids, gather_logits=False), dev, world, warm=2, iters=20)
    finally:
        tp_fused = bool(tp.fused)
        del tp
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    w1 = L * (4 * H * H + 3 * H * I)
    T = B * S
    return {"batch": B, "seq_len": S, "tp_degree": world, "ms": round(dt * 1e3, 1), "ms_min": round(dmin * 1e3, 1), "iterations": 20,
            "tokens_per_s": round(T / dt, 1), "onebit_layer_TFLOPs_equivalent": round(2.0 * T * w1 / dt / 1e12, 1),
            "exchanges_per_layer": 2 if world > 1 else 0,
            "bytes_per_exchange_per_rank": (T * H * 4 + T * H * 2) if world > 1 else 0,
            "attention": "onebit_attention_prefill on the local heads", "logits": "own token rows only (not gathered)",
            "glue": "fused row kernels (onebit_rows_qkv_rope_stats / onebit_rows_res_ln_rms / onebit_rows_swiglu_stats)" if tp_fused else "torch ops"}


def measure_eval(model, dev, windows=8, seqlen=2048, requests=32):
    """The reference's evaluation callers at their real shape on this model through the fused prefill
    route: `perplexity` over `windows` windows of `seqlen` tokens (evaluation/lm_eval.py:93-128: one prefill per
    window, fp16 logits, shifted cross-entropy) and `loglikelihood_tokens` over `requests` ragged requests of 40-400
    tokens in one right-padded batch (evaluation/lm_eval/models_utils.py:275-330).  Synthetic token stream."""
    from onebit_amd.evaluate import loglikelihood_tokens, perplexity
    cfg = model.config
    seqlen = min(seqlen, cfg.max_position_embeddings)
    g = torch.Generator(device="cpu").manual_seed(11)
    stream = torch.randint(0, cfg.vocab_size, (1, windows * seqlen), generator=g)
    lens = torch.randint(40, 401, (requests,), generator=g).tolist()
    reqs = []
    for n in lens:
        toks = torch.randint(1, cfg.vocab_size, (n,), generator=g).tolist()
        ncont = max(1, min(8, n // 8))
        reqs.append((toks[:-ncont], toks[-ncont:]))
    layer = model.model.layers[0]
    projs = {"q/k/v/o": layer.self_attn.q_proj, "gate/up": layer.mlp.gate_proj, "down": layer.mlp.down_proj}

    def routes(T):
        return {k: ("LDS-DMA GEMM on producer-scaled rows" if p.prescaled_ok(T) else "register-staged MFMA GEMM") for k, p in projs.items()}
    model.set_attention("hip").set_fused_glue(True)
    try:
        with torch.no_grad():
            model(stream[:, :128].to(dev))                                           # warm-up (lazy initialisation)
            perplexity(model, stream[:, :seqlen], seqlen, logits_dtype=torch.float16)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ppl = perplexity(model, stream, seqlen, logits_dtype=torch.float16)
            torch.cuda.synchronize(dev)
            dt_p = time.perf_counter() - t0
            loglikelihood_tokens(model, reqs[:4], batch_size=requests, max_length=seqlen, ragged=False)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            res_pad = loglikelihood_tokens(model, reqs, batch_size=requests, max_length=seqlen, ragged=False)
            torch.cuda.synchronize(dev)
            dt_pad = time.perf_counter() - t0
            # the default route: one onebit_mixed_step over the REAL token rows of the batch, lm_head on the continuation rows
            loglikelihood_tokens(model, reqs, batch_size=requests, max_length=seqlen)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            res = loglikelihood_tokens(model, reqs, batch_size=requests, max_length=seqlen)
            torch.cuda.synchronize(dev)
            dt_l = time.perf_counter() - t0
    finally:
        model.set_attention("eager").set_fused_glue(False)
        torch.cuda.empty_cache()
    pad_to = max(len(c) + len(t) - 1 for c, t in reqs)
    real_rows = sum(len(c) + len(t) - 1 for c, t in reqs)
    return {"perplexity": {"windows": windows, "seqlen": seqlen, "ms_per_window": round(dt_p / windows * 1e3, 2),
                           "tokens_per_s": round(windows * seqlen / dt_p, 1), "ppl_synthetic": round(ppl, 1),
                           "gemm_routes": routes(seqlen), "logits": "fp16 [1, %d, %d] per window, loss in fp16 (lm_eval.py:104-121)" % (seqlen, cfg.vocab_size)},
            "loglikelihood_tokens": {"requests": requests, "lengths": "%d..%d tokens" % (min(lens), max(lens)), "padded_batch": [requests, pad_to],
                                     "real_rows": real_rows, "ms": round(dt_l * 1e3, 1), "tokens_per_s_real": round(real_rows / dt_l, 1),
                                     "route": "ragged: ONE onebit_mixed_step over the real token rows (no right-padding), lm_head on the continuation rows",
                                     "padded_route": {"ms": round(dt_pad * 1e3, 1), "tokens_per_s_padded": round(requests * pad_to / dt_pad, 1),
                                                      "tokens_per_s_real": round(real_rows / dt_pad, 1), "gemm_routes": routes(requests * pad_to),
                                                      "route": "the reference's right-padded [B, S] batch through the module path with the fused glue"},
                                     "max_abs_ll_difference_between_routes": round(max(abs(a[0] - b[0]) for a, b in zip(res, res_pad)), 4),
                                     "finite": bool(all(v[0] == v[0] for v in res)),
                                     "includes": "log_softmax over the continuation rows, gather + greedy flags on the device, sum(contlen) floats to the host (round 6; the reference ships [B, S, vocab] log-probabilities to the host, models_utils.py:331)"},
            "route": "set_fused_glue + onebit_attention_prefill (the config-3 prefill route)", "per": "GPU", "data": "synthetic"}


def measure_cpu_baseline(cfg):
    """The oracle's reference-style CPU path (dense +-1 matrix rebuilt on every call, then a dense
    fp32 GEMV, *g, LayerNorm -- bitnet.py:98-118 restated in C), single thread, on the 7 projections
    of ONE decoder layer; tokens/s extrapolated x num_layers (glue ops and lm_head not included,
    which flatters the CPU)."""
    import numpy as np
    from oracle.oracle import COracle
    c = COracle()
    rng = np.random.default_rng(0)
    H, I = cfg.hidden_size, cfg.intermediate_size
    shapes = [(H, H)] * 4 + [(H, I)] * 2 + [(I, H)]
    scratch = np.empty(max(k * n for k, n in shapes), np.float32)
    layers = []
    for (K, N) in shapes:
        packed = rng.integers(0, 256, (N, K // 8), dtype=np.uint8).view(np.int8)
        x = rng.standard_normal((1, K)).astype(np.float32)
        h = (0.1 * (0.5 + rng.random(K))).astype(np.float32)
        g = (0.1 * (0.5 + rng.random(N))).astype(np.float32)
        layers.append((K, N, packed, x, h, g))
    def c_layer(threads):
        t0 = time.perf_counter()
        for (K, N, packed, x, h, g) in layers:
            c.forward_f32_unpack_every_call(packed, x, h, g, scratch[: K * N].reshape(N, K), threads=threads)
        return time.perf_counter() - t0

    total, nlayers = 0.0, 0
    while nlayers < cfg.num_hidden_layers and total < 8.0:         # bounded sample: <= one token, ~8 s
        total += c_layer(1)
        nlayers += 1
    total /= nlayers
    single = {"value": round(1.0 / (total * cfg.num_hidden_layers), 5), "unit": "tokens/s", "cores": 1, "kind": "port",
              "sample": "%d decoder layers' worth of 1-bit projections, %.3f s per layer" % (nlayers, total)}
    # SURVEY.md 8(d) asks for the host cores: the same C restatement with its output rows dealt to OpenMP threads
    # (each thread still rebuilds its rows of the dense matrix on every call, as the reference does); the thread
    # count is swept and the best one is the headline, with the count stated
    ncpu = os.cpu_count() or 1
    sweep, best = {}, (1, total)
    for th in sorted({t for t in (2, 4, 8, 16, 32, 64, 128, ncpu) if 1 < t <= ncpu}):
        c_layer(th)                                                # spin the team up
        t = min(c_layer(th) for _ in range(3))
        sweep[str(th)] = round(1.0 / (t * cfg.num_hidden_layers), 4)
        if t < best[1]:
            best = (th, t)
    sweep["1"] = single["value"]
    out = {"value": round(1.0 / (best[1] * cfg.num_hidden_layers), 5), "unit": "tokens/s", "cores": best[0], "kind": "port",
           "sample": "one decoder layer's 7 BitLinearInf calls (T=1) through the C restatement of the reference's "
                     "unpack-every-call forward (bitnet.py:98-118), OpenMP over output rows, best of a thread sweep "
                     "(best of 3 per count), %.4f s per layer at %d threads; extrapolated to %d layers (glue ops and "
                     "lm_head not included)" % (best[1], best[0], cfg.num_hidden_layers),
           "host_cpus": ncpu, "thread_sweep_tokens_per_s": sweep, "single_core": single}
    # the reference's own op sequence (ATen ops, bitnet.py:98-118) on ALL host cores, and
    # the same with the dense matrix unpacked once and kept (oracle/oracle.py, torch-ops port)
    try:
        from oracle.oracle import torch_forward_ref_style, torch_unpack_ref_style
        ncpu = os.cpu_count() or 1
        old = torch.get_num_threads()
        torch.set_num_threads(ncpu)
        tl = [(torch.from_numpy(p), torch.from_numpy(x), torch.from_numpy(h), torch.from_numpy(g)) for (_, _, p, x, h, g) in layers]
        def one_layer(dense=None):
            t0 = time.perf_counter()
            for i, (p, x, h, g) in enumerate(tl):
                torch_forward_ref_style(p, x, h, g, dense=None if dense is None else dense[i])
            return time.perf_counter() - t0
        one_layer()                                         # warm the thread pool / allocator
        ts, spent = [], 0.0
        while len(ts) < 3 and spent < 10.0:
            t = one_layer(); ts.append(t); spent += t
        t_all = min(ts)
        dense = [torch_unpack_ref_style(p, torch.float32) for (p, _, _, _) in tl]
        one_layer(dense)
        t_once = min(one_layer(dense) for _ in range(5))
        torch.set_num_threads(old)
        out["all_cores"] = {"value": round(1.0 / (t_all * cfg.num_hidden_layers), 5), "unit": "tokens/s", "cores": ncpu,
                            "kind": "port (the reference's ATen op sequence: int64 broadcast unpack on every call, F.linear, *g, LayerNorm)",
                            "sample": "one decoder layer's 7 projections, best of %d, %.3f s per layer, x %d layers" % (len(ts), t_all, cfg.num_hidden_layers)}
        out["unpack_once"] = {"value": round(1.0 / (t_once * cfg.num_hidden_layers), 5), "unit": "tokens/s", "cores": ncpu,
                              "kind": "port, dense fp32 +-1 matrix unpacked once and cached (what the reference does NOT do)",
                              "sample": "one decoder layer's 7 projections, best of 5, %.4f s per layer, x %d layers" % (t_once, cfg.num_hidden_layers)}
    except Exception as e:                                   # the single-core figure must survive
        out["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def measure_train_layer(dev, T=4096, K=4096, N=11008, iters=5):
    """forward + backward of the train-mode BitLinear (latent weights, SignSTE) at a 7B MLP shape, fp16:
    three GEMMs of 2 T K N flops (z = a S^T, ga = gz S, gS = gz^T a) + the LayerNorm / column-sum kernels around them."""
    import torch
    from onebit_amd.train import BitLinear
    torch.manual_seed(0)
    m = BitLinear(K, N, dtype=torch.float16).to(dev)
    with torch.no_grad():
        m.weight.normal_(0, 0.02)
        m.weight_scale.uniform_(0.05, 0.15)
        m.input_factor.uniform_(0.05, 0.15)
    x = torch.randn(T, K, device=dev, dtype=torch.float16, requires_grad=True)
    gy = torch.randn(T, N, device=dev, dtype=torch.float16)

    def step():
        m(x).backward(gy)
        m.zero_grad(set_to_none=True)
        x.grad = None
    for _ in range(2):
        step()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    return {"layer": "%d->%d" % (K, N), "tokens": T, "dtype": "f16", "ms_forward_backward": round(ms, 3), "TFLOPs": round(6.0 * T * K * N / ms / 1e9, 1),
            "flops": "3 GEMMs x 2 T K N", "kernels": "ob_tgemm128_f16_kernel (sign / x*h applied while staging, STE in the epilogue) + LayerNorm forward / backward + column sums",
            "per": "GPU", "data": "synthetic"}


class Emitter:
    """The one JSON line, printed exactly once: by main() when every leg is done, or by the deadline timer with the legs finished
    so far (`incomplete` names the rest) -- after which every rank leaves with os._exit, whatever its main thread is stuck in."""

    def __init__(self, rank, deadline_s):
        self.rank, self.out, self.done, self.pending = rank, None, False, []
        self.lock = threading.Lock()
        self.timer = threading.Timer(deadline_s, self._fire)
        self.timer.daemon = True
        self.deadline_s = deadline_s

    def arm(self, out):
        self.out = out
        self.timer.start()

    def emit(self, reason=None):
        with self.lock:
            if self.done or self.out is None:
                return False
            self.done = True
            if reason:
                self.out["incomplete"] = {"reason": reason, "legs_not_finished": list(self.pending)}
            sys.stdout.flush()
            print(json.dumps(self.out), flush=True)       # the one JSON line, last on stdout
            return True

    def _fire(self):
        if self.rank == 0:
            self.emit("deadline of %.0f s after the headline measurement reached with a secondary leg still running" % self.deadline_s)
        else:
            time.sleep(2.0)                               # (rank 0 prints first)
        sys.stdout.flush()
        os._exit(0)

    def cancel(self):
        self.timer.cancel()


class Hooks:
    """Everything main() touches besides its own control flow (which ranks run what, where the collectives and
    barriers sit, how the JSON line is assembled).  The world-2 gloo test (tests/test_bench_cpu.py) substitutes CPU
    stand-ins, so the control flow the driver launches on N GPUs has executed with more than one rank."""
    backend = "nccl"

    def device(self, local_rank):
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        return dev

    def init_process_group(self, dev, timeout_s=120.0):
        import datetime
        import torch.distributed as dist
        # a collective that does not complete within the timeout must not take the process (and the JSON line) with it: the
        # watchdog's abort is switched off, the hung leg is ended by bench.py's own deadline (Emitter below)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        dist.init_process_group(self.backend, device_id=dev, timeout=datetime.timedelta(seconds=timeout_s))

    def sync(self, dev):
        torch.cuda.synchronize(dev)

    def empty_cache(self):
        torch.cuda.empty_cache()

    def load_lib(self):
        from onebit_amd import _lib
        _lib.load()

    def build_model(self, cfg, seed, dev):
        from onebit_amd.llama import build_synthetic_model
        return build_synthetic_model(cfg, seed=seed, device=dev)

    def make_stepper(self, model, max_len):
        from onebit_amd.engine import DecodeEngine
        return DecodeEngine(model, max_len=max_len)

    measure_roofline = staticmethod(measure_roofline)
    measure_prefill_sharded = staticmethod(measure_prefill_sharded)
    measure_continuous_batch = staticmethod(measure_continuous_batch)
    measure_mixed_step = staticmethod(measure_mixed_step)
    measure_decode_ctx = staticmethod(measure_decode_ctx)
    measure_prefill_model = staticmethod(measure_prefill_model)
    measure_ttft = staticmethod(measure_ttft)
    measure_prefill_model_tp = staticmethod(measure_prefill_model_tp)
    measure_k_sharded_decode = staticmethod(measure_k_sharded_decode)
    measure_cpu_baseline = staticmethod(measure_cpu_baseline)
    measure_eval = staticmethod(measure_eval)
    measure_train_layer = staticmethod(measure_train_layer)


def main(argv=None, hooks=None):
    args = parse(argv)
    hk = hooks or Hooks()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain process: launch the N ranks ourselves (their rank 0 prints the JSON line)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(spawn_command(args.gpus, sys.argv[1:] if argv is None else argv), env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher provides WORLD_SIZE {world}; refusing to report a "
                         "rank count that was not measured")
    rccl_ranks = 1
    dev = hk.device(local_rank)
    if world > 1:
        import torch.distributed as dist
        hk.init_process_group(dev, args.pg_timeout)
        rccl_ranks = dist.get_world_size()
        assert rccl_ranks == args.gpus, (rccl_ranks, args.gpus)

    hk.load_lib()
    cfg = model_config(args.model)
    model = hk.build_model(cfg, 1000 * rank, dev)
    g = torch.Generator(device="cpu").manual_seed(rank)
    prompt = torch.randint(0, cfg.vocab_size, (1, args.prompt), generator=g).to(dev)
    total_new = args.warmup + args.steps + 1

    engine = args.engine
    stepper = None
    if engine in ("auto", "fused"):
        try:
            stepper = hk.make_stepper(model, args.prompt + total_new + 1)
            engine = "fused"
        except (ImportError, ValueError) as e:
            if engine == "fused":
                raise
            print(f"bench: fused engine unavailable ({e}); using the module path", file=sys.stderr)
            engine = "eager"
    if stepper is None:
        cache = model.new_cache(1, args.prompt + total_new + 1)
        logits = model(prompt, cache)
        state = {"tok": logits[:, -1].argmax(-1, keepdim=True)}

        def step():
            lg = model(state["tok"], cache)
            state["tok"] = lg[:, -1].argmax(-1, keepdim=True)
    else:
        stepper.prefill(prompt)
        step = stepper.step

    def fence():
        hk.sync(dev)
        if world > 1:
            dist.barrier()
            hk.sync(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- the headline line exists from here on; every further leg only ADDS a field to it ----
    em = Emitter(rank, args.deadline)
    out = None
    if rank == 0:
        ctx = args.prompt + args.warmup + args.steps // 2
        star_b, tok_b = token_bytes(cfg, ctx)
        tok_s = world * args.steps / dt
        per_gpu_tok_s = args.steps / dt
        out = {
            "metric": "decode_tokens_per_sec", "value": round(tok_s, 2), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "LLaMA-%s OneBit greedy decode, batch=1 per GPU, prompt %d, synthetic "
                                   "inference checkpoint (reference layout), whole token incl. attention + lm_head"
                                   % (args.model.upper(), args.prompt),
                       "engine": engine, "replicas": world, "rccl_ranks": rccl_ranks,
                       "launcher": "torch.distributed.run (one rank per GPU)" if world > 1 else "single process"},
            "token_hbm": {"algorithmic_bytes_per_token": tok_b, "onebit_layer_bytes_per_token": star_b,
                          "achieved_GBps_whole_token": round(tok_b * per_gpu_tok_s / 1e9, 1),
                          "frac_of_8TBps": round(tok_b * per_gpu_tok_s / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": None, "cpu_baseline": None, "prefill_k_sharded": None,
        }
    em.arm(out)

    def leg(name, fn, where="rank0", into=None, key=None):
        """Run one secondary measurement: an exception becomes {"error": ...}, a hang is ended by the Emitter's deadline; the result
        goes to out[name] (or into[key]) on rank 0.  `where`: "rank0" (the other ranks skip it) or "all" (every rank enters)."""
        if where == "rank0" and rank != 0:
            return None
        em.pending.append(name)
        try:
            res = fn()
        except Exception as e:              # the decode line must survive a failure of a secondary measurement
            res = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0 and res is not None:
            (out if into is None else into)[name if key is None else key] = res
        em.pending.remove(name)
        return res

    if not args.no_roofline:                    # right after the decode phase: same thermal / clock state as the headline
        ctx0 = args.prompt + args.warmup + args.steps // 2
        leg("roofline", lambda: hk.measure_roofline(model, dev, dt / args.steps * 1e3, token_bytes(cfg, ctx0)[1]))
    if not args.no_prefill:
        leg("prefill_k_sharded", lambda: hk.measure_prefill_sharded(cfg, dev, world, rank), where="all")
    serve = None
    if not args.no_serve:
        serve = leg("continuous_batch", lambda: hk.measure_continuous_batch(model, dev))
    if not args.no_prefill:
        leg("prefill_model", lambda: hk.measure_prefill_model(model, dev))
        if hasattr(hk, "measure_ttft"):
            leg("time_to_first_token", lambda: hk.measure_ttft(model, dev))
    if not args.no_eval:                        # the evaluation callers at their real shape
        leg("eval_ppl", lambda: hk.measure_eval(model, dev))
    if not args.no_train:                       # the train-mode layer
        leg("train_layer", lambda: hk.measure_train_layer(dev))
    if not args.no_serve and hasattr(hk, "measure_decode_ctx"):      # long-context decode (single stream and 32 slots)
        leg("decode_ctx", lambda: hk.measure_decode_ctx(model, dev))
    if not args.no_prefill:
        def tp_leg():
            nonlocal model
            if world > 1:                       # the same checkpoint on every rank (the decode replicas were seeded per rank)
                model = None
                hk.empty_cache()
                model = hk.build_model(cfg, 4242, dev)
            return hk.measure_prefill_model_tp(model, dev, world, rank)
        leg("prefill_model_tp", tp_leg, where="all")
    if not args.no_k_sharded_decode:            # BASELINE config 4 on LLaMA-13B shapes at every N (N = 1: the line the N > 1 ones compare with)
        def ksd_leg():
            nonlocal model, stepper
            stepper = model = None
            hk.empty_cache()
            res = hk.measure_k_sharded_decode(model_config("13b"), dev, world, rank, min(args.steps, 16), args.prompt)
            res["model"] = "LLaMA-13B shapes"
            return res
        leg("decode_k_sharded", ksd_leg, where="all")
    if serve is not None and "error" not in serve and rank == 0 and args.model == "7b":
        # BASELINE config 5 names LLaMA2-13B: the 32-slot steady-state step and the mixed prefill + decode workload on a 13B-shaped checkpoint
        m13 = {}

        def build13():
            nonlocal model, stepper
            stepper = model = None
            hk.empty_cache()
            m13["m"] = hk.build_model(model_config("13b"), 4242, dev)
            return hk.measure_continuous_batch(m13["m"], dev)
        leg("continuous_batch.llama2_13b_shapes", build13, into=serve, key="llama2_13b_shapes")
        if "m" in m13:
            def mixed13():
                res = hk.measure_mixed_step(m13["m"], dev)
                res["model"] = "LLaMA2-13B shapes"
                return res
            leg("continuous_batch.mixed", mixed13, into=serve, key="mixed")
        m13.clear()
        hk.empty_cache()
    if not args.no_cpu_baseline:                # at every N (rank 0's host cores; the other ranks wait at the barrier below)
        leg("cpu_baseline", lambda: hk.measure_cpu_baseline(cfg))
    if world > 1:
        try:
            dist.barrier()
        except Exception:
            pass
    em.cancel()
    if rank == 0:
        em.emit()                               # BEFORE the process group is torn down: a stuck communicator cannot eat the line
    if world > 1:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
