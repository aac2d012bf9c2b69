"""LLaMA-13B layer widths (hidden 5120, intermediate 13824, 40 heads of 128: BASELINE configs 4 and 5) against the REFERENCE model:
2 layers, vocabulary 512, logits recorded from the reference's ``BitLlamaForCausalLMInf`` (modeling_bitllama.py:1512-1611) on the
CPU of the build container by ``tests/golden/gen_goldens_model_wide.py e`` (fp16 and fp32 parameters; 12-token prompt + 3 decode
steps; 8 sequences batched + 2 batched decode steps).  Weights regenerate bit-exactly from the seed.

At these widths the decode launches take geometries no 7B test reaches (one projection per workgroup with 80 workgroups per q / k / v
and 124 per gate / up, K = 13824 in three vector rounds) and config 4's K-sharded module path runs on them -- until round 4 both were
compared with this repo's own module path only (tests/test_gpu_config4.py).  Bar: ``max(2 x the reference's own fp16-vs-fp32 gap,
2e-3 x logit scale)``, as for the other model tests.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wide13(golden_dir):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM, synthetic_state_dict
    z = np.load(os.path.join(golden_dir, "model_wide_e.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    cfg = OneBitLlamaConfig(**kw)
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.head_dim, cfg.num_attention_heads) == (5120, 13824, 128, 40)
    sd = synthetic_state_dict(cfg, seed=int(z["seed"]), dtype=torch.float16, device="cpu")

    def build():
        m = OneBitLlamaForCausalLM(cfg, torch.float16)
        m.load_state_dict(sd)
        return m.to(torch.device("cuda:0")).eval()
    return z, cfg, build


def _tol(z, name):
    ref16, ref32 = z[name + "_f16"], z[name + "_f32"]
    return __import__('_parity_log').loose_tol(ref16, ref32)


def _check(got, ref16, ref32, tol, what, name="logits"):
    from _parity_log import check
    check("model_wide_e", name, what, got, ref16, ref32, loose=tol)


@pytest.mark.parametrize("route", ["module", "fused+hip", "k-sharded world 1"])
def test_prefill_and_decode_routes_13b_width(wide13, route):
    from onebit_amd.sharded import KShardedBitLinear, shard_model_k
    z, cfg, build = wide13
    dev = torch.device("cuda:0")
    model = build()
    if route == "k-sharded world 1":            # what bench.py's decode_k_sharded runs on every rank (config 4)
        shard_model_k(model, 0, 1, mode="allreduce", copy=False)
        assert sum(isinstance(m, KShardedBitLinear) for m in model.modules()) == 14
    elif route == "fused+hip":
        model.set_attention("hip")
        model.set_fused_glue(True)
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    toks = torch.from_numpy(z["greedy_f16"]).to(dev)
    cache = model.new_cache(1, 32)
    lg = model(ids, cache).cpu().numpy()
    _check(lg, z["prefill_logits_f16"], z["prefill_logits_f32"], _tol(z, "prefill_logits"), route, "prefill_logits")
    assert int(lg[0, -1].argmax()) == int(toks[0, 0])
    n = z["decode_logits_f16"].shape[1]
    dec = np.concatenate([model(toks[:, i:i + 1], cache).cpu().numpy() for i in range(n)], axis=1)
    _check(dec, z["decode_logits_f16"], z["decode_logits_f32"], _tol(z, "decode_logits"), route + " decode", "decode_logits")


@pytest.mark.parametrize("use_graph", [True, False])
def test_decode_engine_13b_width(wide13, use_graph):
    from onebit_amd.engine import DecodeEngine
    z, cfg, build = wide13
    dev = torch.device("cuda:0")
    model = build()
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    eng = DecodeEngine(model, max_len=32, use_graph=use_graph)
    eng.prefill(ids)
    toks = z["greedy_f16"][0]
    assert eng.first_token == int(toks[0])
    tol = _tol(z, "decode_logits")
    for i in range(z["decode_logits_f16"].shape[1]):
        eng.set_state(int(toks[i]), ids.shape[1] + i)            # teacher-forced with the reference's tokens
        eng.step()
        lg = eng.logits().cpu().numpy()
        _check(lg, z["decode_logits_f16"][0, i], z["decode_logits_f32"][0, i], tol, f"DecodeEngine graph={int(use_graph)} step {i}", "decode_logits")
        assert int(lg.argmax()) == int(toks[i + 1])


def test_batched_decode_step_13b_width(wide13):
    from onebit_amd.engine import BatchedDecodeStep
    z, cfg, build = wide13
    dev = torch.device("cuda:0")
    model = build()
    bids = torch.from_numpy(z["batch_ids"]).to(dev)
    B, S = bids.shape
    max_len = 16
    cache = model.new_cache(B, max_len)
    lg = model(bids, cache)[:, -1].cpu().numpy()
    ref16, ref32 = z["batch_logits_f16"], z["batch_logits_f32"]
    tol = __import__('_parity_log').loose_tol(ref16, ref32)
    _check(lg, ref16[:, 0], ref32[:, 0], tol, "module path, batched prefill", "batch_prefill")
    step = BatchedDecodeStep(model, cache.layers, B, max_len, sample=True, keep_logits=True)
    toks = z["batch_greedy_f16"]
    for i in range(ref16.shape[1] - 1):
        step.tokens.copy_(torch.from_numpy(toks[:, i].astype(np.int32)))
        step.pos.fill_(S + i)
        step.launch()
        torch.cuda.synchronize()
        got = step.logits.float().cpu().numpy()
        _check(got, ref16[:, 1 + i], ref32[:, 1 + i], tol, f"BatchedDecodeStep step {i}", "batch_decode")
