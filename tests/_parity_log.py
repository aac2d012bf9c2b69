"""Model-level parity bookkeeping: every route's measured max |delta logit| against the reference's fp16 AND fp32 goldens is
written to gpurun_out/r06_model_parity.txt (copied to profiles/) when OB_WRITE_PROFILES=1.

The BAR is derived from the golden itself, not from this implementation's own output (round-5 review, weak #1 / advisor):

    bar = max(GAP_FACTOR x gap + half an fp16 ulp at the logit scale, 2e-3 x logit scale),   gap = max |reference fp16 logits - reference fp32 logits|

``gap`` is what the reference's OWN fp16 arithmetic costs on this input: the maximum over the logits of one realisation of fp16
rounding noise through the same network.  A correct fp16 implementation with a different (equally valid) summation order is
another, independent realisation of that noise: its distance to the fp32 golden is distributed like ``gap`` itself, its distance
to the fp16 golden like the difference of two independent realisations, i.e. sqrt(2) x as wide.  GAP_FACTOR = 1.6 = sqrt(2) x
1.13: the sqrt(2) of that difference plus 13 % for the spread of a maximum over 10^4 - 10^6 logits from box to box (the module
path's torch GEMMs are not bit-reproducible across boxes).  Measured ratios error / gap over rounds 4-6, all routes, 8 boxes:
0.70 ... 1.30 (``OBSERVED`` below) -- inside the bar with >= 23 % to spare, while a real defect (a dropped rounding point, a
wrong LayerNorm statistic) moves the error by integer multiples of the gap.  ``OBSERVED`` is kept as a LOGGED regression
indicator (the profile line says how the run compares with it); it is no longer asserted."""
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


GAP_FACTOR = 1.6


def loose_tol(ref16, ref32):
    """GAP_FACTOR x gap + half an fp16 ulp at the logit scale (the logits of every route are fp16 tensors: their own output rounding is
    +- half an ulp whatever the gap is -- at 13B widths the reference's gap on a 4-logit slice is 0.0054 and an fp16 ulp 0.0039, so
    errors come in steps of 0.0039 and the gap term alone would put the bar between one and two steps), never below 2e-3 x scale."""
    scale = float(np.abs(ref32).max())
    half_ulp = 0.5 * 2.0 ** (np.floor(np.log2(max(scale, 1e-6))) - 10)
    return max(GAP_FACTOR * float(np.abs(ref16 - ref32).max()) + half_ulp, 2e-3 * scale)


def check(fixture, name, route, got, ref16, ref32, loose=None):
    """assert `got` against both goldens; returns (err16, err32)."""
    ref16 = ref16 if ref16.shape == got.shape else ref16.reshape(got.shape)
    ref32 = ref32 if ref32.shape == got.shape else ref32.reshape(got.shape)
    got = got.astype(np.float32)
    e16, e32 = float(np.abs(got - ref16).max()), float(np.abs(got - ref32).max())
    gap, scale = float(np.abs(ref16 - ref32).max()), float(np.abs(ref32).max())
    bar = loose_tol(ref16, ref32) if loose is None else loose          # (callers that compare a slice pass the whole golden's bar)
    obs = OBSERVED.get((fixture, name))
    if os.environ.get("OB_WRITE_PROFILES") == "1":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        note = ""
        if obs:
            note = "  | worst of rounds 4-5: %.5f / %.5f%s" % (obs[0], obs[1], "  ABOVE IT" if e16 > obs[0] or e32 > obs[1] else "")
        with open(os.path.join(ROOT, "gpurun_out", "r06_model_parity.txt"), "a") as f:
            f.write(f"{fixture:<18s} {name:<16s} {route:<34s} vs ref fp16 {e16:.5f} ({e16 / max(gap, 1e-12):.2f} x gap)  vs ref fp32 {e32:.5f} "
                    f"({e32 / max(gap, 1e-12):.2f} x gap)  | reference's own fp16-fp32 gap {gap:.5f}  logit scale {scale:.3f}  bar {bar:.5f}{note}\n")
    assert e16 <= bar, (fixture, name, route, "fp16 golden", e16, bar)
    assert e32 <= bar, (fixture, name, route, "fp32 golden", e32, bar)
    return e16, e32
