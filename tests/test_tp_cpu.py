"""World-size-2 gloo test of the model-level tensor-parallel prefill (onebit_amd/tp.py): q/k/v
N-sharded by head -> local attention -> o K-sharded (one reduce_scatter), gate/up N-sharded -> down
K-sharded (one reduce_scatter), against the prefill logits recorded from the reference model
(tests/golden/model_tiny_b.npz).  Compute callbacks are oracle-backed (test infrastructure); on a
GPU box the same control flow runs with the HIP callbacks over RCCL."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_sharded_cpu import _free_port, _np_epilogue, _np_partial, _np_rows_u


def _load(golden_dir, name, dtype):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), dtype)
    sd = {k[3:]: (torch.from_numpy(z[k]) if z[k].dtype == np.int8 else torch.from_numpy(z[k]).to(dtype))
          for k in z.files if k.startswith("sd_")}
    model.load_state_dict(sd)
    return z, model.eval()


class _TorchGlue:
    """Torch stand-ins with the contracts of onebit_amd.tp.HipGlue (the fused row kernels), reference op order: the gloo
    tests run tp.py's fused control flow -- the one bench.py launches on N GPUs -- on the CPU."""

    @staticmethod
    def _ln(u, mean, rstd):
        return ((u.float() - mean[:, None]) * rstd[:, None]).to(u.dtype)

    @staticmethod
    def qkv_rope(u_q, u_k, u_v, st6, cos, sin, B, S, Hl, Hkvl, D, eps):
        from onebit_amd.llama import _rotate_half
        ln = _TorchGlue._ln
        q = ln(u_q, st6[:, 0], st6[:, 1]).view(B, S, Hl, D)
        k = ln(u_k, st6[:, 2], st6[:, 3]).view(B, S, Hkvl, D).transpose(1, 2)
        v = ln(u_v, st6[:, 4], st6[:, 5]).view(B, S, Hkvl, D).transpose(1, 2)
        c, s_ = cos[:S].to(q.dtype), sin[:S].to(q.dtype)
        q = q * c[None, :, None] + _rotate_half(q) * s_[None, :, None]           # modeling_bitllama.py:175-181
        k = k * c[None, None] + _rotate_half(k) * s_[None, None]
        return q.contiguous(), k.contiguous(), v.contiguous()

    @staticmethod
    def swiglu(u_g, u_u, st4, eps):
        ln = _TorchGlue._ln
        return torch.nn.functional.silu(ln(u_g, st4[:, 0], st4[:, 1])) * ln(u_u, st4[:, 2], st4[:, 3])

    @staticmethod
    def u_rows(shard, z, dtype):
        g = shard.weight_scale.to(dtype)
        return (z.to(dtype) * g) if dtype != torch.float16 else (z.half() * g)

    @staticmethod
    def res_ln_rms(h, u, w, rms_eps, ln_eps):
        y = torch.nn.functional.layer_norm(u.float(), (u.shape[1],), eps=ln_eps).to(u.dtype)
        hn = h + y
        var = hn.float().pow(2).mean(-1, keepdim=True)
        return hn, w * (hn.float() * torch.rsqrt(var + rms_eps)).to(hn.dtype)

    @staticmethod
    def attention(q, k, v):
        B, S, Hl, D = q.shape
        if k.shape[1] != Hl:
            k, v = k.repeat_interleave(Hl // k.shape[1], dim=1), v.repeat_interleave(Hl // v.shape[1], dim=1)
        w = torch.matmul(q.transpose(1, 2), k.transpose(2, 3)) / (D ** 0.5)
        if S > 1:
            w = w + torch.triu(torch.full((S, S), torch.finfo(w.dtype).min, dtype=w.dtype), diagonal=1)[None, None]
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        return torch.matmul(w, v).transpose(1, 2).reshape(B * S, Hl * D)


def _wall_timed(fn, dev, world, warm=1, iters=2):
    """bench._timed's contract on a CPU: (median_s, min_s) of the slowest rank."""
    import time
    for _ in range(1):
        fn()
    if world > 1:
        dist.barrier()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    t = torch.tensor([sorted(ts)[len(ts) // 2], min(ts)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0]), float(t[1])


def _worker_fused(rank, world, port, golden_dir, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from onebit_amd.sharded import _torch_normalize, _torch_row_stats
        from onebit_amd.tp import TensorParallelPrefill
        z, model = _load(golden_dir, "b", torch.float32)
        kw = dict(rows_fn=_np_rows_u, stats_fn=_torch_row_stats, normalize_fn=_torch_normalize, partial_fn=_np_partial,
                  epilogue_fn=_np_epilogue, glue=_TorchGlue)
        tp = TensorParallelPrefill(model, rank, world, **kw)
        assert tp.fused
        ids = torch.from_numpy(z["input_ids"])
        ref = z["prefill_logits_f32"]
        err = float(np.abs(tp(ids).numpy() - ref).max())
        ids2 = torch.cat([ids[:, :5], ids[:, 3:8]], dim=0)                     # ragged: T = 10 rows, padded per rank
        unfused = TensorParallelPrefill(model, rank, world, **{k: v for k, v in kw.items() if k != "glue"})
        assert not unfused.fused
        err2 = float((tp(ids2) - unfused(ids2)).abs().max())
        ex = tp.exchanges
        # bench.py's own measurement function, as the driver launches it on N ranks: same arguments, stand-in compute
        res = bench.measure_prefill_model_tp(model, torch.device("cpu"), world, rank, B=2, S=8, tp_kwargs=kw, timed=_wall_timed)
        errs = [None] * world
        dist.all_gather_object(errs, (err, err2, ex, float(np.abs(ref).max()), res))
        if rank == 0:
            out.put((errs, model.config.num_hidden_layers, model.config.hidden_size))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_fused_tensor_parallel_flow_and_bench_measure_gloo(golden_dir, world):
    """The fused tensor-parallel control flow (what `bench.py --gpus N` runs as `prefill_model_tp`) with more than one
    rank: logits against the reference's recorded prefill logits, against the unfused flow on a ragged batch, and
    bench.measure_prefill_model_tp itself end to end (its JSON fields)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fused, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    errs, L, H = q.get(timeout=5)
    for err, err2, exchanges, scale, res in errs:
        assert exchanges == 2 * L
        assert err <= 2e-3 * max(1.0, scale), (err, scale)
        assert err2 <= 2e-3 * max(1.0, scale), err2
        assert res["tp_degree"] == world and res["batch"] == 2 and res["seq_len"] == 8 and res["exchanges_per_layer"] == 2
        assert res["bytes_per_exchange_per_rank"] == 16 * H * 6 and res["tokens_per_s"] > 0 and "fused" in res["glue"]


def _worker(rank, world, port, golden_dir, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from onebit_amd.sharded import _torch_normalize, _torch_row_stats
        from onebit_amd.tp import TensorParallelPrefill
        z, model = _load(golden_dir, "b", torch.float32)
        tp = TensorParallelPrefill(model, rank, world, rows_fn=_np_rows_u, stats_fn=_torch_row_stats,
                                   normalize_fn=_torch_normalize, partial_fn=_np_partial, epilogue_fn=_np_epilogue)
        cfg = model.config
        # the plan splits heads and intermediate columns without overlap or gap
        plans = [None] * world
        dist.all_gather_object(plans, (list(tp.plan.heads), list(tp.plan.inter)[:1] + list(tp.plan.inter)[-1:]))
        ids = torch.from_numpy(z["input_ids"])                     # [1, 8]
        logits = tp(ids)
        ref = z["prefill_logits_f32"]
        err = float(np.abs(logits.numpy() - ref).max())
        # ragged: 2 sequences of 5 tokens (T = 10 is not a multiple of world 3 either; rows are padded)
        ids2 = torch.cat([ids[:, :5], ids[:, 3:8]], dim=0)
        l2 = tp(ids2)
        # reference for the ragged batch: the same code at tensor-parallel degree 1 (no collectives; its
        # [1, 8] prefill is pinned by the reference's goldens above) -- the unsharded module path is GPU-only
        tp1 = TensorParallelPrefill(model, 0, 1, rows_fn=_np_rows_u, stats_fn=_torch_row_stats, normalize_fn=_torch_normalize,
                                    partial_fn=_np_partial, epilogue_fn=_np_epilogue)
        ref2 = tp1(ids2).numpy()
        err1 = float(np.abs(tp1(ids).numpy() - ref).max())
        assert err1 <= 2e-3 * max(1.0, float(np.abs(ref).max())), err1
        err2 = float(np.abs(l2.numpy() - ref2).max())
        errs = [None] * world
        dist.all_gather_object(errs, (err, err2, tp.exchanges, float(np.abs(ref).max())))
        if rank == 0:
            out.put((errs, plans, cfg.num_attention_heads, cfg.intermediate_size, cfg.num_hidden_layers))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_tensor_parallel_prefill_gloo(golden_dir, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    errs, plans, H, I, L = q.get(timeout=5)
    heads = sorted(h for pl in plans for h in pl[0])
    assert heads == list(range(H))
    assert plans[0][1][0] == 0 and plans[-1][1][1] == I - 1 and all(a[1][1] + 1 == b[1][0] for a, b in zip(plans, plans[1:]))
    for err, err2, exchanges, scale in errs:
        assert exchanges == 2 * L                                  # two activation exchanges per layer, not seven
        assert err <= 2e-3 * max(1.0, scale), (err, scale)         # vs the reference's recorded fp32 prefill logits
        assert err2 <= 2e-3 * max(1.0, scale), err2                # vs the unsharded model on a ragged batch


def test_plan_rejects_indivisible_heads():
    from onebit_amd.llama import OneBitLlamaConfig
    from onebit_amd.tp import TPPlan
    cfg = OneBitLlamaConfig(vocab_size=64, hidden_size=256, intermediate_size=704, num_hidden_layers=1,
                            num_attention_heads=4, max_position_embeddings=32)
    with pytest.raises(ValueError):
        TPPlan.make(cfg, 0, 3)
    p = [TPPlan.make(cfg, r, 4) for r in range(4)]
    assert [len(x.heads) for x in p] == [1, 1, 1, 1]
    assert p[0].inter.start == 0 and p[-1].inter.stop == 704 and all(a.inter.stop == b.inter.start for a, b in zip(p, p[1:]))
    assert all(x.inter.start % 32 == 0 for x in p)
