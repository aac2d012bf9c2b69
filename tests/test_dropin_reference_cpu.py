"""Drop-in boundary inside the REFERENCE's own model class (SURVEY.md section 8 row b):
``BitLlamaForCausalLMInf`` (modeling_bitllama.py:1512-1611) built around ``onebit_amd.BitLinearInf`` by the documented
monkey-patch of the import at modeling_bitllama.py:25 -- the seven call sites per layer (:229-231, :451-454) then
construct THIS repo's module.  Checked: module count, state-dict keys / shapes / dtypes equal to the unpatched reference
model, a reference state dict loads cleanly, ``.half()`` leaves ``weight`` int8 and casts only the floating parameters
(modeling_utils.py:696 semantics), the converter's in-place ``.data`` assignments (convert_llama_to_infer_ckpt.py:32-34)
take, and forward on CPU tensors refuses loudly (no CPU fallback).

Build-container only: skipped where /root/reference does not exist (the GPU box)."""
import importlib.metadata as md
import os
import sys

import pytest
import torch

REF_SRC = "/root/reference/transformers/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference fork is only present in the build container")

KW = dict(vocab_size=320, hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
          max_position_embeddings=64)


@pytest.fixture(scope="module")
def ref_modules():
    """The reference fork imported read-only (two dependency pins shimmed as in SURVEY.md appendix A)."""
    orig = md.version
    fake = {"tokenizers": "0.14.1", "huggingface-hub": "0.17.3", "huggingface_hub": "0.17.3"}
    md.version = lambda n: fake.get(n, orig(n))
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_SRC)
    saved = {k: v for k, v in sys.modules.items() if k == "transformers" or k.startswith("transformers.")}
    for k in saved:
        del sys.modules[k]
    try:
        import transformers.models.bitnet as ref_bitnet
        import transformers.models.bitllama.modeling_bitllama as mb
        from transformers import BitLlamaConfig
        yield ref_bitnet, mb, BitLlamaConfig
    finally:
        md.version = orig
        sys.path.remove(REF_SRC)
        for k in [k for k in sys.modules if k == "transformers" or k.startswith("transformers.")]:
            del sys.modules[k]
        sys.modules.update(saved)


@pytest.mark.parametrize("attention_bias", [False, True])
def test_reference_model_built_around_our_module(ref_modules, attention_bias):
    import onebit_amd
    ref_bitnet, mb, BitLlamaConfig = ref_modules
    cfg = BitLlamaConfig(attention_bias=attention_bias, **KW)
    torch.manual_seed(0)
    ref_model = mb.BitLlamaForCausalLMInf(cfg)
    ref_cls = mb.BitLinearInf
    assert ref_cls is ref_bitnet.BitLinearInf
    try:
        ref_bitnet.BitLinearInf = mb.BitLinearInf = onebit_amd.BitLinearInf      # INTEGRATION.md, the monkey-patch form
        model = mb.BitLlamaForCausalLMInf(cfg)
    finally:
        ref_bitnet.BitLinearInf = mb.BitLinearInf = ref_cls
    ours = [m for m in model.modules() if isinstance(m, onebit_amd.BitLinearInf)]
    theirs = [m for m in ref_model.modules() if isinstance(m, ref_cls)]
    assert len(ours) == len(theirs) == 7 * KW["num_hidden_layers"]
    assert not any(isinstance(m, ref_cls) for m in model.modules())
    # identical state-dict contract
    sd_ref, sd = ref_model.state_dict(), model.state_dict()
    assert list(sd) == list(sd_ref)
    for k in sd:
        assert sd[k].shape == sd_ref[k].shape and sd[k].dtype == sd_ref[k].dtype, k
    # a (random) reference checkpoint loads by key
    g = torch.Generator().manual_seed(1)
    for k, v in sd_ref.items():
        if v.dtype == torch.int8:
            v.copy_(torch.randint(-128, 128, v.shape, generator=g, dtype=torch.int8))
        elif v.is_floating_point():
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    missing, unexpected = model.load_state_dict(sd_ref, strict=True)
    assert not missing and not unexpected
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd_ref[k]), k
    for m in ours:
        assert all(not p.requires_grad for p in m.parameters())
        assert (m.bias is not None) == (attention_bias and m.out_features in (KW["hidden_size"],) and m in
                                       [x for l in model.model.layers for x in (l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj, l.self_attn.o_proj)])
    # .half(): floating parameters cast, packed weight stays int8 (modeling_utils.py:696 does the same at load)
    model.half()
    for m in ours:
        assert m.weight.dtype == torch.int8
        assert m.weight_scale.dtype == m.input_factor.dtype == torch.float16
        assert m.bias is None or m.bias.dtype == torch.float16
    # the converter's in-place assignments (convert_llama_to_infer_ckpt.py:32-34)
    m0 = ours[0]
    w_new = torch.randint(-128, 128, m0.weight.shape, generator=g, dtype=torch.int8)
    m0.weight.data = w_new
    m0.weight_scale.data = torch.ones_like(m0.weight_scale) * 0.5
    m0.input_factor.data = torch.ones_like(m0.input_factor) * 2.0
    assert torch.equal(model.state_dict()[next(k for k in sd if k.endswith("q_proj.weight"))], w_new)
    # no CPU fallback: the product path refuses CPU tensors instead of computing something else
    with pytest.raises((RuntimeError, ValueError)):
        model(torch.tensor([[1, 2, 3]]))
