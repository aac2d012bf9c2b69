#!/usr/bin/env python3
"""Golden vectors for the evaluation callers, from the REAL reference code
(build container only; PYTHONDONTWRITEBYTECODE=1, the reference tree is read-only).

* log-likelihoods: the reference's own ``BaseLM._loglikelihood_tokens``
  (evaluation/lm_eval/models_utils.py, imported by path) driving the reference's own
  ``BitLlamaForCausalLMInf`` (tiny config "a", the synthetic checkpoint of model_tiny_a.npz) through
  a minimal ``BaseLM`` subclass whose ``_model_call`` is LMClass's (``self.model(inps)["logits"]``).
* perplexity: ``evaluation/lm_eval.py`` executes at import (hard-coded paths, loggers), so its PPL
  loop (:93-128) is driven here statement by statement against the reference model; the arithmetic
  recorded is the reference model's, the loop is the part under test in ``onebit_amd.evaluate``.

Output: tests/golden/eval_tiny_a.npz (requests, expected log-likelihoods / greedy flags, token
stream, expected PPL for fp32 and fp16 parameters).
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens_eval.py
"""
import importlib.metadata as md
import importlib.util
import os
import sys

import numpy as np
import torch
from torch import nn

_orig = md.version
_fake = {"tokenizers": "0.14.1", "huggingface-hub": "0.17.3", "huggingface_hub": "0.17.3"}
md.version = lambda n: _fake.get(n, _orig(n))
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present; fixtures can only be regenerated in the build container")
sys.path.insert(0, os.path.join(REF, "transformers/src"))
from transformers import BitLlamaConfig, BitLlamaForCausalLMInf  # noqa: E402  (the reference fork)

spec = importlib.util.spec_from_file_location("ref_models_utils", os.path.join(REF, "evaluation/lm_eval/models_utils.py"))
ref_mu = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_mu)

OUT = os.path.dirname(os.path.abspath(__file__))
z = np.load(os.path.join(OUT, "model_tiny_a.npz"))
kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
sd16 = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}


def ref_model(dt):
    m = BitLlamaForCausalLMInf(BitLlamaConfig(**kw))
    m.load_state_dict({k: (v if v.dtype == torch.int8 else v.to(dt)) for k, v in sd16.items()})
    return m.to(dt).eval() if dt == torch.float32 else m.half().eval()


class TinyLM(ref_mu.BaseLM):
    def __init__(self, model, batch_size, max_length):
        super().__init__()
        self.model, self._bs, self._ml = model, batch_size, max_length
    eot_token_id = property(lambda self: 0)
    max_length = property(lambda self: self._ml)
    max_gen_toks = property(lambda self: 8)
    batch_size = property(lambda self: self._bs)
    device = property(lambda self: torch.device("cpu"))
    def tok_encode(self, s): raise NotImplementedError
    def tok_decode(self, t): raise NotImplementedError
    def _model_generate(self, c, m, e): raise NotImplementedError
    def _model_call(self, inps):
        with torch.no_grad():
            return self.model(inps)["logits"]


g = torch.Generator().manual_seed(2024)
V = kw["vocab_size"]
reqs = []
for (lc, ln) in [(5, 3), (1, 1), (12, 7), (3, 20), (9, 1), (30, 10), (2, 2), (7, 5), (16, 16), (4, 9)]:
    reqs.append((torch.randint(1, V, (lc,), generator=g).tolist(), torch.randint(1, V, (ln,), generator=g).tolist()))
reqs.append((reqs[0][0] + reqs[0][1][:1], reqs[0][1][1:]))      # same tokens as request 0, different split
reqs.append((list(reqs[2][0]), list(reqs[2][1])))               # exact duplicate of request 2
with torch.no_grad():                                            # a continuation greedy decoding reproduces
    _ctx = torch.randint(1, V, (1, 6), generator=g)
    _gen = ref_model(torch.float32).generate(_ctx, max_new_tokens=3, do_sample=False)
reqs.append((_ctx[0].tolist(), _gen[0, 6:].tolist()))
BATCH, MAXLEN = 4, 32                                            # request 5 (40 tokens) is left-truncated
out = {"batch_size": np.array(BATCH), "max_length": np.array(MAXLEN), "n_req": np.array(len(reqs))}
for i, (c, t) in enumerate(reqs):
    out["ctx_%d" % i], out["cont_%d" % i] = np.array(c), np.array(t)

stream = torch.randint(0, V, (1, 100), generator=g)              # 3 windows of 32 + a dropped tail
SEQLEN = 32
out["ppl_tokens"], out["ppl_seqlen"] = stream.numpy(), np.array(SEQLEN)

for dt, dn in ((torch.float32, "f32"), (torch.float16, "f16")):
    model = ref_model(dt)
    lm = TinyLM(model, BATCH, MAXLEN)
    ans = lm._loglikelihood_tokens([(None, c, t) for c, t in reqs], disable_tqdm=True)
    out["ll_" + dn] = np.array([a[0] for a in ans], dtype=np.float64)
    out["greedy_" + dn] = np.array([a[1] for a in ans])
    # lm_eval.py:93-128, llama branch, for limit = -1 and limit = 1
    for limit in (-1, 1):
        nsamples = stream.numel() // SEQLEN
        nlls = []
        with torch.no_grad():
            for i in range(nsamples):
                batch = stream[:, (i * SEQLEN):((i + 1) * SEQLEN)]
                hidden_states = model.model(batch)[0]
                logits = model.lm_head(hidden_states)
                shift_logits = logits[:, :-1, :]
                shift_labels = stream[:, (i * SEQLEN):((i + 1) * SEQLEN)][:, 1:]
                loss = nn.CrossEntropyLoss()(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1))
                nlls.append(loss.float() * SEQLEN)
                if i == limit:
                    break
        ppl = torch.exp(torch.stack(nlls).sum() / (nsamples * SEQLEN))
        out["ppl_%s_limit%s" % (dn, "all" if limit < 0 else str(limit))] = np.array(ppl.item())
    print(dn, "ll", out["ll_" + dn][:4], "greedy", out["greedy_" + dn], "ppl", out["ppl_%s_limitall" % dn], out["ppl_%s_limit1" % dn])

np.savez_compressed(os.path.join(OUT, "eval_tiny_a.npz"), **out)
print("wrote eval_tiny_a.npz")
