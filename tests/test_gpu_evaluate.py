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


def test_loglikelihood_tokens_matches_reference(golden_dir):
    from onebit_amd.evaluate import loglikelihood_tokens
    e, model = _load(golden_dir, torch.device("cuda:0"))
    n = int(e["n_req"])
    reqs = [(e["ctx_%d" % i].tolist(), e["cont_%d" % i].tolist()) for i in range(n)]
    got = loglikelihood_tokens(model, reqs, int(e["batch_size"]), int(e["max_length"]))
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
