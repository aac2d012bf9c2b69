"""onebit_attention_prefill (csrc/ob_flash.h: causal flash attention on MFMA, the build's own kernel on the fused
prefill route) against the reference's attention arithmetic (modeling_bitllama.py:546-563: q k^T / sqrt(D) + causal
mask, softmax in fp32, probabilities . v) evaluated in fp32 on the same fp16 inputs."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, kc, vc, past):
    """q [B, S, H, D], caches [B, Hkv, max_len, D] -> [B, S, H, D] in fp32 math."""
    B, S, H, D = q.shape
    L = past + S
    k, v = kc[:B, :, :L].float(), vc[:B, :, :L].float()
    if k.shape[1] != H:
        k, v = k.repeat_interleave(H // k.shape[1], dim=1), v.repeat_interleave(H // v.shape[1], dim=1)
    w = torch.matmul(q.float().transpose(1, 2), k.transpose(2, 3)) / math.sqrt(D)
    mask = torch.triu(torch.full((S, L), float("-inf"), device=q.device), diagonal=past + 1)
    w = torch.softmax(w + mask[None, None], dim=-1)
    return torch.matmul(w, v).transpose(1, 2)


@pytest.mark.parametrize("B,S,H,Hkv,D,past,max_len", [
    (1, 128, 4, 4, 128, 0, 128),        # one full query block
    (2, 200, 8, 2, 64, 0, 256),         # grouped-query, head_dim 64, ragged tail (200 = 128 + 72)
    (1, 77, 4, 4, 128, 50, 160),        # a chunk appended to a non-empty cache (chunked prefill)
    (3, 1, 2, 2, 128, 9, 16),           # a single new token per sequence
    (1, 1000, 2, 1, 128, 0, 1024),      # many key blocks, rescaling of the running output
    (2, 33, 4, 4, 64, 31, 64),          # the diagonal inside a key block
])
def test_against_reference_arithmetic(B, S, H, Hkv, D, past, max_len):
    from onebit_amd.llama import hip_attention_prefill
    g = torch.Generator().manual_seed(B * 1000 + S)
    q = torch.randn(B, S, H, D, generator=g).half().to(DEV)
    kc = torch.randn(B, Hkv, max_len, D, generator=g).half().to(DEV)
    vc = torch.randn(B, Hkv, max_len, D, generator=g).half().to(DEV)
    kc[:, :, past + S:] = float("nan")                       # rows beyond the valid keys must never be used
    vc[:, :, past + S:] = float("nan")
    o = hip_attention_prefill(q, kc, vc, past)
    ref = _ref(q, kc, vc, past)
    assert o.shape == q.shape and torch.isfinite(o).all()
    err = float((o.float() - ref).abs().max())
    assert err <= 3e-3 * max(1.0, float(ref.abs().max())), err
    # the same with o_proj's input scaling fused: fp16(fp16(o) * h)
    h = (0.1 * (0.5 + torch.rand(H * D, generator=g))).half().to(DEV)
    o2 = hip_attention_prefill(q, kc, vc, past, h)
    assert torch.equal(o2, (o.view(B, S, H * D) * h).view(B, S, H, D))


def test_large_scores_and_peaked_softmax():
    """Online softmax stability: scores of +-60 (a few keys carry all the mass), and a sequence whose first keys dominate."""
    from onebit_amd.llama import hip_attention_prefill
    B, S, H, D = 1, 300, 2, 128
    g = torch.Generator().manual_seed(5)
    q = (6.0 * torch.randn(B, S, H, D, generator=g)).half().to(DEV)
    kc = (1.0 * torch.randn(B, H, 320, D, generator=g)).half().to(DEV)
    kc[:, :, :3] *= 4.0
    vc = torch.randn(B, H, 320, D, generator=g).half().to(DEV)
    o = hip_attention_prefill(q, kc, vc, 0)
    ref = _ref(q, kc, vc, 0)
    assert torch.isfinite(o).all()
    assert float((o.float() - ref).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("per_block_log2", [5.0, 9.0, 0.5])
def test_slowly_and_quickly_rising_maximum(per_block_log2):
    """The kernel moves its running maximum only when a 64-key tile exceeds it by more than 2^8 (csrc/ob_flash.h): scores that rise
    by 5 log2 units per key block alternate between the deferred and the rescaling branch, 9 per block rescales every block, 0.5
    defers for 16 blocks in a row -- each against the fp32 arithmetic, with V chosen so that a missed rescale shows."""
    from onebit_amd.llama import hip_attention_prefill
    B, S, H, D = 1, 1024, 2, 128
    g = torch.Generator().manual_seed(11)
    q = (0.05 * torch.randn(B, S, H, D, generator=g)).half()
    kc = (0.05 * torch.randn(B, H, S, D, generator=g)).half()
    slope = per_block_log2 / 64.0 / 1.4426950408889634 * math.sqrt(D)          # raw score units per key
    q[..., 0] = 4.0
    kc[..., 0] = (slope / 4.0) * torch.arange(S, dtype=torch.float32)[None, None, :]
    vc = torch.randn(B, H, S, D, generator=g).half()
    vc[..., 1] = torch.arange(S, dtype=torch.float32)[None, None, :] / 64.0          # a column that tracks WHICH keys carried the mass
    q, kc, vc = q.to(DEV), kc.to(DEV), vc.to(DEV)
    o = hip_attention_prefill(q, kc, vc, 0)
    ref = _ref(q, kc, vc, 0)
    assert torch.isfinite(o).all()
    assert float((o.float() - ref).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max()))


def test_config3_shape_matches_sdpa_and_is_deterministic():
    """BASELINE config 3 attention shape (8 x 2048 tokens, 32 heads of 128): against torch's fused attention on the same
    inputs, bit-identical on repetition."""
    from onebit_amd.llama import hip_attention_prefill
    B, S, H, D = 8, 2048, 32, 128
    g = torch.Generator(device=DEV).manual_seed(1)
    q = torch.randn(B, S, H, D, generator=g, device=DEV, dtype=torch.float16)
    kc = torch.randn(B, H, S, D, generator=g, device=DEV, dtype=torch.float16)
    vc = torch.randn(B, H, S, D, generator=g, device=DEV, dtype=torch.float16)
    o = hip_attention_prefill(q, kc, vc, 0)
    ref = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), kc, vc, is_causal=True).transpose(1, 2)
    assert float((o.float() - ref.float()).abs().max()) <= 4e-3 * max(1.0, float(ref.float().abs().max()))
    assert torch.equal(o, hip_attention_prefill(q, kc, vc, 0))


def test_argument_errors():
    from onebit_amd import _lib
    lib = _lib.load()
    assert lib.onebit_attention_prefill(None, None, None, None, None, 1, 8, 4, 4, 96, 0, 8, None) == -2      # head_dim
    assert lib.onebit_attention_prefill(None, None, None, None, None, 1, 8, 4, 3, 128, 0, 8, None) == -2     # heads % kv heads
    assert lib.onebit_attention_prefill(None, None, None, None, None, 1, 8, 4, 4, 128, 4, 8, None) == -2     # beyond the cache
    assert lib.onebit_attention_prefill(None, None, None, None, None, 1, 8, 4, 4, 128, 0, 8, None) == -1     # null pointers
    assert lib.onebit_attention_prefill(None, None, None, None, None, 0, 8, 4, 4, 128, 0, 8, None) == 0      # empty batch
