"""GPU model-level parity: the host LLaMA around BitLinearInf against logits recorded from the
reference's BitLlamaForCausalLMInf on tiny configs (tests/golden/gen_goldens_model.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load_model(golden_dir, name, dev):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, f"model_tiny_{name}.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    cfg = OneBitLlamaConfig(**kw)
    model = OneBitLlamaForCausalLM(cfg, torch.float16)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    model.load_state_dict(sd)              # reference key layout loads unchanged (strict)
    return z, cfg, model.to(dev).eval()


@pytest.mark.parametrize("name", ["a", "b"])
def test_eager_model_matches_reference_logits(golden_dir, name):
    dev = torch.device("cuda:0")
    z, cfg, model = _load_model(golden_dir, name, dev)
    ids = torch.from_numpy(z["input_ids"]).to(dev)
    cache = model.new_cache(1, 32)
    logits = model(ids, cache).cpu().numpy()
    ref16, ref32 = z["prefill_logits_f16"], z["prefill_logits_f32"]
    assert logits.shape == ref16.shape and logits.dtype == np.float32
    scale = np.abs(ref32).max()
    # the reference's own fp16-vs-fp32 gap sets the scale of what "parity" can mean here
    ref_gap = np.abs(ref16 - ref32).max()
    assert np.abs(logits - ref16).max() <= max(2.0 * ref_gap, 2e-3 * scale), (np.abs(logits - ref16).max(), ref_gap)
    assert np.abs(logits - ref32).max() <= max(2.0 * ref_gap, 2e-3 * scale)
    # incremental decode with the KV cache, feeding the reference's greedy tokens
    toks = torch.from_numpy(z["greedy_f16"]).to(dev)
    dec = []
    for i in range(4):
        dec.append(model(toks[:, i:i + 1], cache).cpu().numpy())
    dec = np.concatenate(dec, axis=1)
    assert np.abs(dec - z["decode_logits_f16"]).max() <= max(2.0 * ref_gap, 2e-3 * scale)
    # greedy tokens agree wherever the reference's own top-2 margin is not within noise
    out = model.generate(ids, max_new_tokens=5)[:, ids.shape[1]:].cpu().numpy()
    margin = z["margin_f16"]
    for i in range(5):
        if margin[:i + 1].min() > 4.0 * max(ref_gap, 1e-3):
            assert out[0, i] == z["greedy_f16"][0, i], (i, out, z["greedy_f16"])
