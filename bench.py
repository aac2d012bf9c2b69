#!/usr/bin/env python3
"""Benchmark: LLaMA-7B OneBit greedy decode (BASELINE.json configs[1]) on N MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one decoded token (batch 1) through the whole model: 224 packed 1-bit projections
(the hot path) plus RMSNorm / RoPE / attention over the KV cache / SiLU / residuals / fp16
lm_head / argmax.  Weights are a synthetic 7B-shaped OneBit inference checkpoint in the
reference's layout, resident in HBM before timing.  For N > 1 every rank decodes its own
sequence on its own full replica (decode does not shard, BASELINE.json north_star: "decode
stays single-GPU") -- weak scaling, no data-path collective; `value` is the aggregate.

The JSON line also carries
  roofline     -- the dominant kernel (the 4096->11008 1-bit GEMV) timed with HIP events over
                  >= 64 distinct weight sets (> 256 MB Infinity Cache), algorithmic bytes / time
                  against the 8 TB/s HBM peak
  cpu_baseline -- the oracle's reference-style CPU path (unpack every call) timed on this host
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def parse():
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
    ap.add_argument("--k-sharded-decode", action="store_true",
                    help="also time BASELINE config 4: module-path decode with every 1-bit layer K-sharded over the "
                         "ranks (one all-reduce per BitLinearInf call); extra JSON field, not the headline value")
    return ap.parse_args()


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


def measure_roofline(model, dev):
    """Dominant kernel of the decode step: the fused gate+up 1-bit GEMV launch (hidden ->
    2 x intermediate, T = 1: residual + LayerNorm + RMSNorm prologue, two projections), exactly the
    launch onebit_decode_step issues.  One launch per decoder layer over that layer's own weights
    (7B: 32 distinct sets = 361 MB, beyond the 256 MB Infinity Cache), captured in a HIP graph and
    replayed; HIP events on the replay stream.  The per-launch time therefore includes the
    dependent-kernel boundary (~1.3 us) that every launch of a decode chain pays; rocprofv3's
    kernel-only duration is in profiles/."""
    from onebit_amd.engine import PRO_RES_LN_RMS, fused_gemv
    cfg = model.config
    K, N = cfg.hidden_size, cfg.intermediate_size
    f16 = torch.float16
    hres = torch.randn(K, device=dev).to(f16)
    u_prev = torch.randn(K, device=dev).to(f16)
    hres_out = torch.empty(K, device=dev, dtype=f16)
    ug, uu = torch.empty(N, device=dev, dtype=f16), torch.empty(N, device=dev, dtype=f16)
    layers = list(model.model.layers)

    def chain():
        for layer in layers:
            fused_gemv([layer.mlp.gate_proj, layer.mlp.up_proj], [ug, uu], PRO_RES_LN_RMS,
                       rms_eps=cfg.rms_norm_eps, hres_in=hres, u_prev=u_prev, hres_out=hres_out,
                       rms_w=layer.post_attention_layernorm.weight)

    chain()
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        chain()
    for _ in range(8):                      # settle clocks / TLBs on the new buffers before timing
        graph.replay()
    torch.cuda.synchronize(dev)
    reps = max(2, 512 // len(layers))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    n = reps * len(layers)
    us = e0.elapsed_time(e1) * 1e3 / n
    ab = 2 * algorithmic_bytes(1, K, N)
    achieved = ab / (us * 1e-6) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("ob_dec_gemv_gateup_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": "ob_dec_gemv_kernel<KV=%d,MS,aligned,RES_LN_RMS,i8,NPROJ=2> (fused gate+up 1-bit GEMV %d->2x%d, T=1)" % ((K + 4095) // 4096, K, N),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": ab, "avg_launch_us": round(us, 3), "launches": n,
            "distinct_weight_sets": len(layers),
            "note": "HIP events around graph replays of back-to-back launches: includes the kernel boundary"}


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
    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(6):                      # the first calls after the decode phase run at ramping clocks
        k_sharded_forward(shard, x, mode="rs_ag")
    fence()
    n = 12
    t0 = time.perf_counter()
    for _ in range(n):
        k_sharded_forward(shard, x, mode="rs_ag")
    fence()
    dt = (time.perf_counter() - t0) / n
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    res = {"layer": "%d->%d" % (K, N), "tokens": T, "k_shards": world, "exchange": "reduce_scatter(fp32)+all_gather(fp16)",
           "ms_per_call": round(dt * 1e3, 3), "tokens_per_s": round(T / dt, 1),
           "TFLOPs": round(2.0 * T * K * N / dt / 1e12, 1), "mfma_peak_TFLOPs": 2500.0 * world,
           "frac_of_mfma_peak": round(2.0 * T * K * N / dt / 1e12 / (2500.0 * world), 4)}
    # the layout that needs no exchange at all: the packed matrix is only N*K/8 bytes, so every rank
    # keeps all of it and takes T/world tokens (token sharding); reported beside the K-sharded path
    from onebit_amd import BitLinearInf
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data, m.input_factor.data, m.weight_scale.data = W, h, gs
    Tl = T // world
    xl = x[rank * Tl:(rank + 1) * Tl]
    for _ in range(6):
        m(xl)
    fence()
    t0 = time.perf_counter()
    for _ in range(n):
        m(xl)
    fence()
    dt2 = (time.perf_counter() - t0) / n
    if world > 1:
        tmax = torch.tensor([dt2], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt2 = float(tmax.item())
    # N-sharded (output rows split): only the LayerNorm statistics cross ranks (2 all-reduces of [T] fp32)
    try:
        from onebit_amd.sharded import n_sharded_forward, shard_n
        nsh = shard_n(W, h, gs, None, rank, world)
        for _ in range(4):
            n_sharded_forward(nsh, x)
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            n_sharded_forward(nsh, x)
        fence()
        dt3 = (time.perf_counter() - t0) / n
        if world > 1:
            tmax = torch.tensor([dt3], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt3 = float(tmax.item())
        res["n_sharded"] = {"rows_per_rank": nsh.n1 - nsh.n0, "exchange": "all_gather(fp32 [T,2] row statistics); output stays N-sharded",
                            "ms_per_call": round(dt3 * 1e3, 3), "tokens_per_s": round(T / dt3, 1),
                            "TFLOPs": round(2.0 * T * K * N / dt3 / 1e12, 1),
                            "frac_of_mfma_peak": round(2.0 * T * K * N / dt3 / 1e12 / (2500.0 * world), 4)}
    except Exception as e:
        res["n_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}
    res["token_sharded"] = {"tokens_per_rank": Tl, "ms_per_call": round(dt2 * 1e3, 3), "tokens_per_s": round(Tl * world / dt2, 1),
                            "TFLOPs": round(2.0 * Tl * world * K * N / dt2 / 1e12, 1),
                            "frac_of_mfma_peak": round(2.0 * Tl * world * K * N / dt2 / 1e12 / (2500.0 * world), 4)}
    return res


def measure_k_sharded_decode(cfg, dev, world, rank, steps, prompt_len):
    """BASELINE config 4: greedy decode with every BitLinearInf K-sharded (onebit_amd/sharded.py,
    KShardedBitLinear): partial GEMV on the rank's K slice, all-reduce of the [1, N] fp32 partials,
    g + LayerNorm everywhere.  Module path (torch glue, no HIP graph): the point is the exchange."""
    import torch.distributed as dist
    from onebit_amd.llama import build_synthetic_model
    from onebit_amd.sharded import shard_model_k
    model = build_synthetic_model(cfg, seed=4242, device=dev)          # same checkpoint on every rank
    shard_model_k(model, rank, world, mode="allreduce")
    torch.cuda.empty_cache()
    g = torch.Generator(device="cpu").manual_seed(7)
    prompt = torch.randint(0, cfg.vocab_size, (1, prompt_len), generator=g).to(dev)
    cache = model.new_cache(1, prompt_len + steps + 4)
    tok = model(prompt, cache)[:, -1].argmax(-1, keepdim=True)
    for _ in range(2):
        tok = model(tok, cache)[:, -1].argmax(-1, keepdim=True)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tok = model(tok, cache)[:, -1].argmax(-1, keepdim=True)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return {"k_shards": world, "exchange": "all_reduce(fp32 [1,N]) per BitLinearInf call", "steps": steps,
            "ms_per_token": round(dt / steps * 1e3, 3), "tokens_per_s": round(steps / dt, 2),
            "collectives_per_token": 7 * cfg.num_hidden_layers if world > 1 else 0, "path": "module (eager)"}


def measure_continuous_batch(model, dev, slots=32, steps=40):
    """BASELINE config 5 shape on this model: `slots` sequences decoding together through
    onebit_decode_step_batched (one HIP-graph replay per step); steady-state decode steps only."""
    from onebit_amd.serving import ContinuousBatcher
    cfg = model.config
    g = torch.Generator(device="cpu").manual_seed(3)
    cb = ContinuousBatcher(model, max_batch=slots, max_len=16 + steps + 16)
    for _ in range(slots):
        cb.add_request(torch.randint(0, cfg.vocab_size, (16,), generator=g).tolist(), steps + 12)
    for _ in range(6):
        cb.step()                                   # prefill step, graph capture, warm replays
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        cb.step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    return {"slots": slots, "ms_per_step": round(dt * 1e3, 3), "tokens_per_s": round(slots / dt, 1),
            "engine": "onebit_decode_step_batched" if cb._native is not None else "torch glue",
            "note": "steady-state decode, all slots active, greedy; per GPU"}


def measure_prefill_model(model, dev, B=8, S=2048):
    """BASELINE configs[2]: whole-model prefill of B x S tokens (1-bit GEMMs + fused row glue +
    the vendor's fused attention), tokens/s and the 1-bit layers' share expressed in TFLOP/s."""
    cfg = model.config
    S = min(S, cfg.max_position_embeddings)
    ids = torch.randint(0, cfg.vocab_size, (B, S), generator=torch.Generator(device="cpu").manual_seed(5)).to(dev)
    model.set_attention("sdpa").set_fused_glue(True)
    try:
        with torch.no_grad():
            model(ids[:1, :128]); model(ids)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            n = 2
            for _ in range(n):
                model(ids)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / n
    finally:
        model.set_attention("eager").set_fused_glue(False)
        torch.cuda.empty_cache()
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    w1 = L * (4 * H * H + 3 * H * I)
    return {"batch": B, "seq_len": S, "ms": round(dt * 1e3, 1), "tokens_per_s": round(B * S / dt, 1),
            "onebit_layer_TFLOPs_equivalent": round(2.0 * B * S * w1 / dt / 1e12, 1),
            "attention": "sdpa", "glue": "onebit_rows_res_ln_rms + onebit_rows_swiglu", "per": "GPU"}


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
    total, nlayers = 0.0, 0
    while nlayers < cfg.num_hidden_layers and total < 12.0:        # bounded sample: <= one token, ~12 s
        t0 = time.perf_counter()
        for (K, N, packed, x, h, g) in layers:
            c.forward_f32_unpack_every_call(packed, x, h, g, scratch[: K * N].reshape(N, K))
        total += time.perf_counter() - t0
        nlayers += 1
    total /= nlayers
    tok_s = 1.0 / (total * cfg.num_hidden_layers)
    return {"value": round(tok_s, 5), "unit": "tokens/s", "cores": 1, "kind": "port",
            "sample": "%d decoder layers' worth of 1-bit projections (7 BitLinearInf calls each, T=1) through the C "
                      "restatement of the reference's unpack-every-call forward, %.2f s per layer; extrapolated to %d "
                      "layers (glue ops and lm_head not included)" % (nlayers, total, cfg.num_hidden_layers),
            "host_cpus": os.cpu_count()}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from onebit_amd import _lib
    from onebit_amd.llama import build_synthetic_model
    _lib.load()
    cfg = model_config(args.model)
    model = build_synthetic_model(cfg, seed=1000 * rank, device=dev)
    g = torch.Generator(device="cpu").manual_seed(rank)
    prompt = torch.randint(0, cfg.vocab_size, (1, args.prompt), generator=g).to(dev)
    total_new = args.warmup + args.steps + 1

    engine = args.engine
    stepper = None
    if engine in ("auto", "fused"):
        try:
            from onebit_amd.engine import DecodeEngine
            stepper = DecodeEngine(model, max_len=args.prompt + total_new + 1)
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
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

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

    roof = None
    if rank == 0 and not args.no_roofline:      # right after the decode phase: same thermal / clock state as the headline
        roof = measure_roofline(model, dev)
    prefill = None
    if not args.no_prefill:
        try:
            prefill = measure_prefill_sharded(cfg, dev, world, rank)
        except Exception as e:              # the decode line must survive a failure of the secondary measurement
            prefill = {"error": "%s: %s" % (type(e).__name__, e)}
    ksd = None
    if args.k_sharded_decode:
        try:
            ksd = measure_k_sharded_decode(cfg, dev, world, rank, min(args.steps, 16), args.prompt)
        except Exception as e:
            ksd = {"error": "%s: %s" % (type(e).__name__, e)}
    serve = None
    if not args.no_serve and rank == 0:
        try:
            serve = measure_continuous_batch(model, dev)
        except Exception as e:
            serve = {"error": "%s: %s" % (type(e).__name__, e)}
    pmodel = None
    if not args.no_prefill and rank == 0:
        try:
            pmodel = measure_prefill_model(model, dev)
        except Exception as e:
            pmodel = {"error": "%s: %s" % (type(e).__name__, e)}
    cpu = None
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cpu = measure_cpu_baseline(cfg)
    if world > 1:
        dist.barrier()

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
                       "engine": engine, "replicas": world},
            "token_hbm": {"algorithmic_bytes_per_token": tok_b, "onebit_layer_bytes_per_token": star_b,
                          "achieved_GBps_whole_token": round(tok_b * per_gpu_tok_s / 1e9, 1),
                          "frac_of_8TBps": round(tok_b * per_gpu_tok_s / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": roof, "cpu_baseline": cpu, "prefill_k_sharded": prefill,
        }
        if ksd is not None:
            out["decode_k_sharded"] = ksd
        if serve is not None:
            out["continuous_batch"] = serve
        if pmodel is not None:
            out["prefill_model"] = pmodel
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)       # the one JSON line, last on stdout


if __name__ == "__main__":
    main()
