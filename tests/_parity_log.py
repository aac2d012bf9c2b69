"""Model-level parity bookkeeping (round-4 review, weak #1): every route's measured max |delta logit| against the reference's
fp16 AND fp32 goldens is written to gpurun_out/r05_model_parity.txt (copied to profiles/) when OB_WRITE_PROFILES=1, and the
tests' bar is 1.25 x the worst error OBSERVED for that golden (``OBSERVED``, from profiles/r05_model_parity.txt), never
looser than the round-1..4 bar max(2 x the reference's own fp16-vs-fp32 gap, 2e-3 x logit scale)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# worst max |logit - reference logit| over all routes, per golden array, as measured on MI355X (profiles/r05_model_parity.txt).
# key: (fixture file, golden array name) -> (vs the reference's fp16 logits, vs its fp32 logits)
OBSERVED = {
    # maxima over 8 runs on different boxes (the module path's torch GEMMs -- eager attention, batched prefill -- are not bit-reproducible
    # from box to box: a golden's worst error moves by up to 20 % between runs)
    ("model_wide_d", "prefill_logits"): (0.02002, 0.01847),      # 32 layers, 7B widths: the reference's own fp16-fp32 gap is 0.01675
    ("model_wide_d", "decode_logits"): (0.01660, 0.01610),       #   (gap 0.01327)
    ("model_wide_d", "batch_prefill"): (0.01685, 0.01604),       #   (gap 0.01618)
    ("model_wide_d", "batch_decode"): (0.02106, 0.01918),        #   (gap 0.01666)
    ("model_wide_c", "prefill_logits"): (0.00488, 0.00663),      # 2 layers, 7B widths (gap 0.00604)
    ("model_wide_c", "decode_logits"): (0.00488, 0.00676),       #   (gap 0.00632)
    ("model_wide_c", "long_logits"): (0.00635, 0.00671),         #   4096-token prompt (gap 0.00601)
    ("model_wide_c", "batch_prefill"): (0.00490, 0.00742),       #   (gap 0.00669)
    ("model_wide_c", "batch_decode"): (0.00586, 0.00767),        #   (gap 0.00779)
    ("model_wide_c", "kshard_decode"): (0.00488, 0.00625),       #   FusedKShardedDecoder, worlds 3 / 8 in lockstep on one device
    ("model_wide_e", "prefill_logits"): (0.00774, 0.00761),      # 2 layers, 13B widths (gap 0.00761)
    ("model_wide_e", "decode_logits"): (0.00781, 0.00762),       #   (gap 0.00600)
    ("model_wide_e", "batch_prefill"): (0.00586, 0.00766),       #   (gap 0.00651)
    ("model_wide_e", "batch_decode"): (0.00586, 0.00747),        #   (gap 0.00764)
    ("model_wide_e", "kshard_decode"): (0.00781, 0.00769),       #   FusedKShardedDecoder, worlds 1 / 2 / 4 / 8
}


def loose_tol(ref16, ref32):
    return max(2.0 * float(np.abs(ref16 - ref32).max()), 2e-3 * float(np.abs(ref32).max()))


def check(fixture, name, route, got, ref16, ref32, loose=None):
    """assert `got` against both goldens; returns (err16, err32)."""
    ref16 = ref16 if ref16.shape == got.shape else ref16.reshape(got.shape)
    ref32 = ref32 if ref32.shape == got.shape else ref32.reshape(got.shape)
    got = got.astype(np.float32)
    e16, e32 = float(np.abs(got - ref16).max()), float(np.abs(got - ref32).max())
    gap, scale = float(np.abs(ref16 - ref32).max()), float(np.abs(ref32).max())
    loose = loose_tol(ref16, ref32) if loose is None else loose        # (callers that compare a slice pass the whole golden's bar)
    obs = OBSERVED.get((fixture, name))
    t16 = min(loose, 1.25 * obs[0]) if obs else loose
    t32 = min(loose, 1.25 * obs[1]) if obs else loose
    if os.environ.get("OB_WRITE_PROFILES") == "1":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r05_model_parity.txt"), "a") as f:
            f.write(f"{fixture:<18s} {name:<16s} {route:<34s} vs ref fp16 {e16:.5f}  vs ref fp32 {e32:.5f}  | reference's own fp16-fp32 gap {gap:.5f}  "
                    f"logit scale {scale:.3f}  bar {t16:.5f} / {t32:.5f} (round-4 bar {loose:.5f})\n")
    assert e16 <= t16, (fixture, name, route, "fp16 golden", e16, t16)
    assert e32 <= t32, (fixture, name, route, "fp32 golden", e32, t32)
    return e16, e32
