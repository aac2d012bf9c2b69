"""Continuous batching on the GPU: every request's tokens equal ``model.generate`` on that request
alone (up to fp16 near-ties), with more requests than slots and mixed prefill + decode steps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(golden_dir, name, dev):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    return model.to(dev).eval()


@pytest.mark.parametrize("use_graph,native", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("name", ["a", "b"])
def test_continuous_batch_matches_single_sequence_generate(golden_dir, name, use_graph, native):
    from onebit_amd.serving import ContinuousBatcher
    dev = torch.device("cuda:0")
    model = _model(golden_dir, name, dev)
    V = model.config.vocab_size
    g = torch.Generator().manual_seed(5)
    reqs = [(torch.randint(0, V, (n,), generator=g).tolist(), m) for n, m in
            [(8, 6), (1, 9), (13, 3), (5, 1), (20, 7), (2, 12), (9, 5)]]
    cb = ContinuousBatcher(model, max_batch=3, max_len=40, use_graph=use_graph, native=native)
    assert (cb._native is not None) == (native and name == "b")     # config a: intermediate 688 is not a multiple of 32
    rids = [cb.add_request(p, m) for p, m in reqs]
    out = cb.run()
    assert cb.steps < sum(m for _, m in reqs)              # steps were shared between requests
    assert (cb.graph_steps > 0) == use_graph
    for rid, (p, m) in zip(rids, reqs):
        ref = model.generate(torch.tensor([p], device=dev), m)[0, len(p):].tolist()
        got = out[rid]
        assert len(got) == m
        if got != ref:                                     # tolerate only an fp16 near-tie at the first divergence
            j = next(i for i in range(m) if got[i] != ref[i])
            lg = model(torch.tensor([p + ref[:j]], device=dev))[0, -1]
            assert abs(float(lg[got[j]] - lg[ref[j]])) < 2e-2 * float(lg.abs().max()), (rid, j)


def test_native_batched_step_on_7b_shaped_layers():
    """onebit_decode_step_batched with real layer widths (hidden 4096 / intermediate 11008 / head_dim
    128, 2 layers), 5 slots of which one stays idle, requests of different lengths: tokens equal
    single-sequence generate (fp16 near-ties tolerated at the first divergence)."""
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    from onebit_amd.serving import ContinuousBatcher
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=512, hidden_size=4096, intermediate_size=11008, num_hidden_layers=2,
                            num_attention_heads=32, max_position_embeddings=64)
    model = build_synthetic_model(cfg, seed=3, device=dev)
    g = torch.Generator().manual_seed(8)
    reqs = [(torch.randint(0, 512, (n,), generator=g).tolist(), m) for n, m in [(6, 8), (1, 5), (11, 8), (3, 2)]]
    cb = ContinuousBatcher(model, max_batch=5, max_len=32)
    assert cb._native is not None
    rids = [cb.add_request(p, m) for p, m in reqs]
    out = cb.run()
    assert cb.graph_steps > 0
    for rid, (p, m) in zip(rids, reqs):
        ref = model.generate(torch.tensor([p], device=dev), m)[0, len(p):].tolist()
        got = out[rid]
        if got != ref:
            j = next(i for i in range(m) if got[i] != ref[i])
            lg = model(torch.tensor([p + ref[:j]], device=dev))[0, -1]
            assert abs(float(lg[got[j]] - lg[ref[j]])) < 2e-2 * float(lg.abs().max()), (rid, j)


@pytest.mark.parametrize("batch,vocab", [(5, 1000), (32, 32000), (48, 4099)])
def test_batched_lm_head_and_argmax_inside_the_step(batch, vocab):
    """onebit_decode_step_batched with next_tokens: the fp16 logits of every slot against the dense
    product of the step's own final-norm output with lm_head (fp32 accumulate -> fp16, as
    modeling_bitllama.py:1610 computes them), and the greedy token = argmax (first index on ties)."""
    from onebit_amd.engine import BatchedDecodeStep
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=vocab, hidden_size=512, intermediate_size=1408, num_hidden_layers=1,
                            num_attention_heads=8, max_position_embeddings=32)
    model = build_synthetic_model(cfg, seed=21, device=dev)
    shape = (batch, cfg.num_key_value_heads, 16, cfg.head_dim)
    caches = [(torch.zeros(shape, device=dev, dtype=torch.float16), torch.zeros(shape, device=dev, dtype=torch.float16))]
    step = BatchedDecodeStep(model, caches, batch, 16, sample=True, keep_logits=True)
    g = torch.Generator().manual_seed(3)
    step.tokens.copy_(torch.randint(0, vocab, (batch,), generator=g).to(torch.int32))
    step.pos.zero_()
    x = step.launch()
    torch.cuda.synchronize()
    ref = (x.float() @ model.lm_head.weight.float().t())
    got = step.logits.float()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2.0 ** -9 * max(1.0, scale)          # fp16 rounding of the same sums
    tok = step.next_tokens.long()
    assert tok.min() >= 0 and tok.max() < vocab
    # greedy token: argmax of the kernel's own fp16 logits, lowest index on ties
    assert torch.equal(tok, got.argmax(-1)) or all(
        float(got[b, tok[b]]) == float(got[b].max()) and int(tok[b]) == int((got[b] == got[b].max()).nonzero()[0]) for b in range(batch))


@pytest.mark.parametrize("native", [False, True])
def test_chunked_prefill_produces_the_same_tokens(golden_dir, native):
    """prefill_chunk = 4: prompts enter the cache in chunks while other requests decode (chunks with a non-empty
    cache take the masked attention branch); every request's tokens equal the unchunked batcher's."""
    from onebit_amd.serving import ContinuousBatcher
    dev = torch.device("cuda:0")
    model = _model(golden_dir, "b", dev)
    V = model.config.vocab_size
    g = torch.Generator().manual_seed(7)
    reqs = [(torch.randint(0, V, (n,), generator=g).tolist(), m) for n, m in [(11, 5), (3, 8), (17, 4), (1, 6), (8, 3)]]

    def run(chunk):
        cb = ContinuousBatcher(model, max_batch=3, max_len=40, native=native, prefill_chunk=chunk)
        rids = [cb.add_request(p, m) for p, m in reqs]
        out = cb.run()
        return [out[r] for r in rids], cb.steps
    ref, steps_ref = run(None)
    got, steps = run(4)
    assert steps > steps_ref                                   # the long prompts took several steps to enter
    for (p, m), a, b in zip(reqs, got, ref):
        assert len(a) == m
        if a != b:                                             # tolerate only an fp16 near-tie at the first divergence
            j = next(i for i in range(m) if a[i] != b[i])
            lg = model(torch.tensor([p + b[:j]], device=dev))[0, -1]
            assert abs(float(lg[a[j]] - lg[b[j]])) < 2e-2 * float(lg.abs().max())


def test_decode_bursts_give_the_same_tokens(golden_dir):
    """``max_burst`` > 1: while nothing can be admitted and nobody finishes, steps are enqueued back to back with each
    step's tokens fed to the next ON THE DEVICE and read by the host after one synchronisation -- the same tokens, step
    and token accounting as one synchronisation per step (requests of different lengths, more requests than slots)."""
    from onebit_amd.serving import ContinuousBatcher
    dev = torch.device("cuda:0")
    model = _model(golden_dir, "b", dev)
    V = model.config.vocab_size
    g = torch.Generator().manual_seed(9)
    reqs = [(torch.randint(0, V, (n,), generator=g).tolist(), m) for n, m in
            [(6, 14), (3, 9), (11, 20), (2, 5), (7, 17), (4, 11)]]
    outs, steps = {}, {}
    for mb in (1, 8):
        cb = ContinuousBatcher(model, max_batch=4, max_len=48, max_burst=mb)
        assert cb._native is not None
        rids = [cb.add_request(p, m) for p, m in reqs]
        out = cb.run()
        outs[mb] = [out[r] for r in rids]
        steps[mb] = (cb.steps, cb.tokens_scheduled)
        assert all(len(o) == m for o, (_, m) in zip(outs[mb], reqs))
        if mb > 1:
            assert cb._graph_fb is not None                       # bursts really ran
    assert outs[1] == outs[8]
    assert steps[1] == steps[8]
