"""Full-WIDTH model parity against the REFERENCE model, not against this repo's own module path: LLaMA-7B layer
widths (hidden 4096, intermediate 11008, 32 heads of 128; 2 layers, vocabulary 512) -- logits recorded from the
reference's ``BitLlamaForCausalLMInf`` (modeling_bitllama.py:1512-1611) on the CPU of the build container by
tests/golden/gen_goldens_model_wide.py.  The weights are regenerated here from the same seed
(``synthetic_state_dict(cfg, seed, device="cpu")`` is bit-reproducible); the fixture holds ids and logits only.

Every decode / prefill route of the build is held to a bar derived from the golden itself,
``max(1.6 x the reference's own fp16-vs-fp32 gap, 2e-3 x logit scale)`` (``tests/_parity_log.py`` says why 1.6; the errors measured
per route are logged to ``profiles/r06_model_parity.txt``):
  module path (eager), fused glue + fused attention on a 4096-token prompt (the LDS-DMA GEMM on producer-scaled
  rows), ``DecodeEngine`` (HIP graph and direct), ``BatchedDecodeStep`` at 32 slots, ``ContinuousBatcher``.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wide(golden_dir):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM, synthetic_state_dict
    z = np.load(os.path.join(golden_dir, "model_wide_c.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    cfg = OneBitLlamaConfig(**kw)
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.head_dim) == (4096, 11008, 128)
    model = OneBitLlamaForCausalLM(cfg, torch.float16)
    model.load_state_dict(synthetic_state_dict(cfg, seed=int(z["seed"]), dtype=torch.float16, device="cpu"))
    return z, cfg, model.to(torch.device("cuda:0")).eval()


def _tol(z, a="prefill_logits"):
    ref16, ref32 = z[a + "_f16"], z[a + "_f32"]
    return __import__('_parity_log').loose_tol(ref16, ref32)


def _chk(z, name, route, got, sl=None):
    from _parity_log import check
    from _parity_log import loose_tol
    r16, r32 = z[name + "_f16"], z[name + "_f32"]
    loose = max(loose_tol(r16, r32), _tol(z)) if name == "decode_logits" else loose_tol(r16, r32)
    if sl is not None:
        r16, r32 = r16[sl], r32[sl]
    return check("model_wide_c", name, route, got, r16, r32, loose=loose)


def test_module_path_prefill_and_decode(wide):
    z, cfg, model = wide
    dev = torch.device("cuda:0")
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    cache = model.new_cache(1, 32)
    lg = model(ids, cache).cpu().numpy()
    _chk(z, "prefill_logits", "module path", lg)
    toks = torch.from_numpy(z["greedy_f16"]).to(dev)
    assert int(lg[0, -1].argmax()) == int(toks[0, 0])
    dec = np.concatenate([model(toks[:, i:i + 1], cache).cpu().numpy() for i in range(4)], axis=1)
    _chk(z, "decode_logits", "module path", dec)
    # fused glue on the short prompt as well (T = 12: skinny GEMM route, not pre-scaled)
    model.set_fused_glue(True)
    try:
        lgf = model(ids, model.new_cache(1, 32)).cpu().numpy()
    finally:
        model.set_fused_glue(False)
    _chk(z, "prefill_logits", "fused glue, 12 tokens", lgf)


def test_fused_prefill_4096_tokens_lds_dma_gemm(wide):
    """A 4096-token prompt (at T = 2048 only gate / up fill the chip with 256 x 256 tiles; from 4096 every projection
    of this width does): every projection takes the LDS-DMA GEMM on rows the producers pre-scaled
    (ONEBIT_FLAG_PRESCALED; asserted, not assumed), fused q|k|v glue and fused causal attention.  16 positions
    of the reference's logits (the last 8 and 8 seeded ones)."""
    z, cfg, model = wide
    dev = torch.device("cuda:0")
    ids = torch.from_numpy(z["long_ids"]).to(dev)
    T = ids.shape[1]
    for layer in model.model.layers:
        for p in (layer.self_attn.q_proj, layer.self_attn.o_proj, layer.mlp.gate_proj, layer.mlp.down_proj):
            assert p.prescaled_ok(T), "this shape is expected on the LDS-DMA GEMM"
    pos = z["long_pos"]
    tol = _tol(z, "long_logits")
    outs = {}
    for name, fused, attn in (("eager", False, "eager"), ("fused", True, "sdpa"), ("fused+hip", True, "hip")):
        try:
            model.set_attention(attn)
        except ValueError:
            continue                                     # an attention implementation this build does not have
        model.set_fused_glue(fused)
        try:
            lg = model(ids, model.new_cache(1, T))[0, pos].cpu().numpy()
        finally:
            model.set_fused_glue(False)
            model.set_attention("eager")
        outs[name] = lg
        _chk(z, "long_logits", "4096-token prompt, " + name, lg)
    assert "eager" in outs and "fused" in outs


@pytest.mark.parametrize("use_graph", [True, False])
def test_decode_engine(wide, use_graph):
    from onebit_amd.engine import DecodeEngine
    z, cfg, model = wide
    dev = torch.device("cuda:0")
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    eng = DecodeEngine(model, max_len=32, use_graph=use_graph)
    eng.prefill(ids)
    toks = z["greedy_f16"][0]
    assert eng.first_token == int(toks[0])
    for i in range(4):
        eng.set_state(int(toks[i]), ids.shape[1] + i)            # teacher-forced with the reference's tokens
        eng.step()
        lg = eng.logits().cpu().numpy()
        _chk(z, "decode_logits", f"DecodeEngine graph={int(use_graph)} step {i}", lg, sl=(0, i))


@pytest.mark.parametrize("prescaled_rows", [True, False])
def test_batched_decode_step_32_slots(wide, prescaled_rows):
    """(prescaled_rows: the row kernels write fp16(x * input_factor) per consumer and every projection takes the LDS-DMA
    skinny GEMM, ob_skinny3.h; without: the first-form kernels scale x themselves.)  32 sequences: batched prefill through the module path (checked against the reference's batched call), then
    three native batched steps (onebit_decode_step_batched: skinny 1-bit GEMMs, row kernels, per-slot attention,
    batched lm_head) teacher-forced with the reference's tokens -- logits of every slot against the reference."""
    from onebit_amd.engine import BatchedDecodeStep
    z, cfg, model = wide
    dev = torch.device("cuda:0")
    bids = torch.from_numpy(z["batch_ids"]).to(dev)
    B, S = bids.shape
    max_len = 16
    cache = model.new_cache(B, max_len)
    lg = model(bids, cache)[:, -1].cpu().numpy()
    ref16, ref32 = z["batch_logits_f16"], z["batch_logits_f32"]
    tol = __import__('_parity_log').loose_tol(ref16, ref32)
    from _parity_log import check
    check("model_wide_c", "batch_prefill", "module path, 32 sequences", lg, ref16[:, 0], ref32[:, 0], loose=tol)
    step = BatchedDecodeStep(model, cache.layers, B, max_len, sample=True, keep_logits=True, prescaled_rows=prescaled_rows)
    toks = z["batch_greedy_f16"]
    for i in range(3):
        step.tokens.copy_(torch.from_numpy(toks[:, i].astype(np.int32)))
        step.pos.fill_(S + i)
        step.launch()
        torch.cuda.synchronize()
        got = step.logits.float().cpu().numpy()
        check("model_wide_c", "batch_decode", f"BatchedDecodeStep prescaled={int(prescaled_rows)} step {i}", got, ref16[:, 1 + i], ref32[:, 1 + i],
              loose=tol)
        nxt = step.next_tokens.cpu().numpy()
        srt = np.sort(ref16[:, 1 + i], axis=-1)
        clear = (srt[:, -1] - srt[:, -2]) > 2.0 * tol                # the reference's own top-2 margin is not noise
        assert (nxt[clear] == toks[clear, 1 + i]).all()


def test_continuous_batcher_tokens(wide):
    """The scheduler + native step end to end: 32 requests of 6 prompt tokens, 4 new tokens each; every request's
    tokens equal the reference's batched greedy tokens up to the first position whose top-2 margin is within noise."""
    from onebit_amd.serving import ContinuousBatcher
    z, cfg, model = wide
    bids, toks = z["batch_ids"], z["batch_greedy_f16"]
    ref16, ref32 = z["batch_logits_f16"], z["batch_logits_f32"]
    tol = __import__('_parity_log').loose_tol(ref16, ref32)
    cb = ContinuousBatcher(model, max_batch=32, max_len=16)
    assert cb._native is not None
    rids = [cb.add_request(bids[b].tolist(), 4) for b in range(bids.shape[0])]
    out = cb.run()
    assert cb.graph_steps > 0
    srt = np.sort(ref16, axis=-1)
    margin = srt[..., -1] - srt[..., -2]                             # [32, 4]
    for b, rid in enumerate(rids):
        got = out[rid]
        assert len(got) == 4
        for i in range(4):
            if margin[b, i] <= 2.0 * tol:
                break                                                # a near-tie: later tokens may legitimately diverge
            assert got[i] == int(toks[b, i]), (b, i, got, toks[b].tolist())
