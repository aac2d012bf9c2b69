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
