"""Evaluation callers on the GPU model against values recorded from the reference's own
``_loglikelihood_tokens`` / PPL arithmetic on the reference model (tests/golden/gen_goldens_eval.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, dev):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, "model_tiny_a.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float16)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
    return np.load(os.path.join(golden_dir, "eval_tiny_a.npz")), model.to(dev).eval()


@pytest.mark.parametrize("ragged", [None, False])
def test_loglikelihood_tokens_matches_reference(golden_dir, ragged):
    """The reference's padded batch through the module path against the reference harness's recorded values.  (The recorded model
    has intermediate_size 688, not a multiple of 32: the native engines refuse it, so ragged=None falls back to the padded route --
    asserted -- and ragged=True raises; the ragged route is held to the padded one in the test below.)"""
    from onebit_amd import evaluate
    from onebit_amd.evaluate import loglikelihood_tokens
    e, model = _load(golden_dir, torch.device("cuda:0"))
    n = int(e["n_req"])
    reqs = [(e["ctx_%d" % i].tolist(), e["cont_%d" % i].tolist()) for i in range(n)]
    assert evaluate._ragged_scorer(model, 4, 32) is None
    with pytest.raises(RuntimeError):
        loglikelihood_tokens(model, reqs, int(e["batch_size"]), int(e["max_length"]), ragged=True)
    got = loglikelihood_tokens(model, reqs, int(e["batch_size"]), int(e["max_length"]), ragged=ragged)
    ll = np.array([g[0] for g in got])
    # fp16 tolerance: the reference's own fp16-vs-fp32 gap per request, doubled, plus 2e-3 relative
    gap = np.abs(e["ll_f16"] - e["ll_f32"])
    tol = 2.0 * gap + 2e-3 * np.abs(e["ll_f32"]) + 2e-3
    assert (np.abs(ll - e["ll_f16"]) <= tol).all(), (np.abs(ll - e["ll_f16"]), tol)
    assert [g[1] for g in got] == e["greedy_f16"].tolist()
    assert got[0] == got[11 - 1] and got[2] == got[12 - 1]          # grouped requests share one evaluation


def test_perplexity_matches_reference(golden_dir):
    from onebit_amd.evaluate import perplexity
    e, model = _load(golden_dir, torch.device("cuda:0"))
    toks, S = torch.from_numpy(e["ppl_tokens"]), int(e["ppl_seqlen"])
    for lim, key in ((-1, "all"), (1, "1")):
        ref16, ref32 = float(e["ppl_f16_limit" + key]), float(e["ppl_f32_limit" + key])
        got = perplexity(model, toks, S, limit=lim, logits_dtype=torch.float16)
        tol = 2.0 * abs(ref16 - ref32) + 2e-3 * ref32
        assert abs(got - ref16) <= tol, (got, ref16, ref32)


def test_loglikelihood_tokens_ragged_equals_padded_at_width():
    """32 requests of 44 .. 390 tokens (bench.py's eval workload) on a 1024-wide synthetic model: the ragged route (one mixed step of
    the real rows, split at max_rows) against the padded batch -- log-likelihoods within fp16 noise, greedy flags equal wherever the
    padded route's top-2 margin is clear, rank sharding unchanged."""
    from onebit_amd.evaluate import loglikelihood_tokens
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    dev = torch.device("cuda:0")
    cfg = OneBitLlamaConfig(vocab_size=2048, hidden_size=1024, intermediate_size=2816, num_hidden_layers=3, num_attention_heads=8,
                            num_key_value_heads=4, max_position_embeddings=512)
    model = build_synthetic_model(cfg, seed=5, device=dev)
    g = torch.Generator().manual_seed(1)
    reqs = []
    for i in range(32):
        n, c = int(torch.randint(44, 391, (1,), generator=g)), int(torch.randint(1, 9, (1,), generator=g))
        t = torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist()
        reqs.append((t[:-c], t[-c:]))
    reqs.append((reqs[3][0], reqs[3][1]))                       # a duplicate: grouped with request 3
    reqs.append(([5], [7]))                                     # a one-row input
    pad = loglikelihood_tokens(model, reqs, 16, 400, ragged=False)
    for kw in (dict(ragged=True), dict(ragged=True, max_rows=1000), dict()):
        rag = loglikelihood_tokens(model, reqs, 16, 400, **kw)
        a, b = np.array([p[0] for p in pad]), np.array([r[0] for r in rag])
        assert np.abs(a - b).max() <= 2e-2 + 2e-3 * np.abs(a).max(), np.abs(a - b).max()
        assert sum(p[1] != r[1] for p, r in zip(pad, rag)) <= 1
        assert rag[3] == rag[32]
