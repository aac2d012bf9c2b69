"""Host LLaMA decoder around ``BitLinearInf`` -- the caller of the hot path.

Restates the inference model of the reference
(``transformers/src/transformers/models/bitllama/modeling_bitllama.py``):
``BitLlamaForCausalLMInf`` (:1512) -> ``LlamaModelInf`` (:1189) ->
``LlamaDecoderLayerInf`` (:856) -> ``LlamaAttentionInf`` (:431) / ``LlamaMLPInf`` (:223),
``LlamaRMSNorm`` (:67-81), rotary embedding (:87-113, :167-181).  Module and parameter names
match the reference, so an inference checkpoint written by
``scripts/convert_llama_to_infer_ckpt.py`` (``pytorch_model.bin`` keys
``model.layers.{i}.self_attn.q_proj.{weight,weight_scale,input_factor}`` ...) loads with
``load_state_dict`` unchanged.

This file is the eager ("module") form: every 1-bit projection goes through
``BitLinearInf.forward`` -> C ABI; the glue (RMSNorm, RoPE, attention, SiLU, residuals, lm_head)
is plain torch in the reference's op order and rounding points.  The KV cache is preallocated
(the reference grows it with ``torch.cat``, :536-541).  The fused whole-token decode engine lives
in ``onebit_amd/engine.py``.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import nn

from .bitnet import BitLinearInf


@dataclass
class OneBitLlamaConfig:
    """Defaults = LLaMA-7B, configuration_bitllama.py:117-136."""
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: Optional[int] = None
    max_position_embeddings: int = 2048
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    attention_bias: bool = False
    rope_scaling: Optional[dict] = None       # {"type": "linear" | "dynamic", "factor": float > 1}, configuration_bitllama.py:168-187

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_heads")
        if self.rope_scaling is not None:           # the reference's _rope_scaling_validation
            rs = self.rope_scaling
            if not isinstance(rs, dict) or len(rs) != 2:
                raise ValueError(f"`rope_scaling` must be a dictionary with two fields, `type` and `factor`, got {rs}")
            if rs.get("type") not in ("linear", "dynamic"):
                raise ValueError(f"`rope_scaling`'s type field must be one of ['linear', 'dynamic'], got {rs.get('type')}")
            if not isinstance(rs.get("factor"), float) or rs["factor"] <= 1.0:
                raise ValueError(f"`rope_scaling`'s factor field must be a float > 1, got {rs.get('factor')}")

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @staticmethod
    def llama_7b() -> "OneBitLlamaConfig":
        return OneBitLlamaConfig()

    @staticmethod
    def llama_13b() -> "OneBitLlamaConfig":
        return OneBitLlamaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40,
                                 num_attention_heads=40)


class LlamaRMSNorm(nn.Module):
    """modeling_bitllama.py:67-81 (weight frozen)."""

    def __init__(self, hidden_size, eps=1e-6, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        input_dtype = hidden_states.dtype
        hidden_states = hidden_states.to(torch.float32)
        variance = hidden_states.pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.variance_epsilon)
        return self.weight * hidden_states.to(input_dtype)


def rope_tables(head_dim: int, max_pos: int, base: float, device, dtype, seq_len: Optional[int] = None,
                scaling: Optional[dict] = None):
    """cos/sin caches of LlamaRotaryEmbedding (:87-113) for `seq_len` (default max_pos) positions: fp32 tables cast to
    the model dtype.  ``scaling``: the reference's two variants -- "linear" (positions divided by the factor, :125-141)
    and "dynamic" NTK (the base grows once the cached length exceeds max_position_embeddings, :144-165)."""
    n = max_pos if seq_len is None else seq_len
    if scaling is not None and scaling["type"] == "dynamic" and n > max_pos:
        f = float(scaling["factor"])
        base = base * ((f * n / max_pos) - (f - 1)) ** (head_dim / (head_dim - 2))
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32, device=device) / head_dim))
    t = torch.arange(n, dtype=torch.float32, device=device)
    if scaling is not None and scaling["type"] == "linear":
        t = t / float(scaling["factor"])
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class LlamaMLPInf(nn.Module):
    """modeling_bitllama.py:223-259."""

    def __init__(self, config: OneBitLlamaConfig, dtype=None):
        super().__init__()
        self.gate_proj = BitLinearInf(config.hidden_size, config.intermediate_size, bias=False, dtype=dtype)
        self.up_proj = BitLinearInf(config.hidden_size, config.intermediate_size, bias=False, dtype=dtype)
        self.down_proj = BitLinearInf(config.intermediate_size, config.hidden_size, bias=False, dtype=dtype)

    def forward(self, x):
        return self.down_proj(nn.functional.silu(self.gate_proj(x)) * self.up_proj(x))


def hip_attention_prefill(q: torch.Tensor, kc: torch.Tensor, vc: torch.Tensor, past_len: int, h_next: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal attention of S new tokens against cache rows 0 .. past_len + S - 1 through ``onebit_attention_prefill``.
    q [B, S, H, D] fp16 contiguous (token-major), kc / vc [slots >= B, Hkv, max_len, D] contiguous with the new tokens'
    keys / values already written; returns [B, S, H, D] (optionally times ``h_next`` [H * D], rounded once more)."""
    from . import _lib
    from .bitnet import _stream_ptr
    B, S, H, D = q.shape
    if kc.shape[0] < B or kc.shape[3] != D or kc.shape != vc.shape or kc.dtype != q.dtype or not q.is_contiguous():
        raise ValueError("hip_attention_prefill: cache / query geometry mismatch")
    o = torch.empty_like(q)
    with torch.cuda.device(q.device):
        rc = _lib.load().onebit_attention_prefill(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(),
                                                  None if h_next is None else h_next.data_ptr(), B, S, H, kc.shape[1], D, past_len,
                                                  kc.shape[2], _stream_ptr(q.device))
    _lib.check(rc, "onebit_attention_prefill")
    return o


class LlamaAttentionInf(nn.Module):
    """modeling_bitllama.py:431-585, eager attention, preallocated KV cache."""

    def __init__(self, config: OneBitLlamaConfig, dtype=None):
        super().__init__()
        self.config = config
        H, Hkv, D = config.num_attention_heads, config.num_key_value_heads, config.head_dim
        self.num_heads, self.num_key_value_heads, self.head_dim = H, Hkv, D
        self.hidden_size = config.hidden_size
        self.q_proj = BitLinearInf(self.hidden_size, H * D, bias=config.attention_bias, dtype=dtype)
        self.k_proj = BitLinearInf(self.hidden_size, Hkv * D, bias=config.attention_bias, dtype=dtype)
        self.v_proj = BitLinearInf(self.hidden_size, Hkv * D, bias=config.attention_bias, dtype=dtype)
        self.o_proj = BitLinearInf(H * D, self.hidden_size, bias=config.attention_bias, dtype=dtype)
        self.attn_impl = "eager"          # "sdpa": fused kernel for prefill from an empty cache

    def forward(self, hidden_states, cos, sin, kv: Tuple[torch.Tensor, torch.Tensor], past_len: int,
                pre_ln_out: bool = False):
        B, S, _ = hidden_states.shape
        # pre_ln_out: the caller fuses o_proj's LayerNorm (and adds its bias, if any) with what follows (onebit_rows_res_ln_rms_bias)
        o_proj = self.o_proj.pre_layernorm_bias_deferred if pre_ln_out else self.o_proj
        H, Hkv, D = self.num_heads, self.num_key_value_heads, self.head_dim
        q = self.q_proj(hidden_states).view(B, S, H, D).transpose(1, 2)
        k = self.k_proj(hidden_states).view(B, S, Hkv, D).transpose(1, 2)
        v = self.v_proj(hidden_states).view(B, S, Hkv, D).transpose(1, 2)
        c = cos[past_len:past_len + S][None, None]
        s = sin[past_len:past_len + S][None, None]
        q = (q * c) + (_rotate_half(q) * s)                  # apply_rotary_pos_emb, :175-181
        k = (k * c) + (_rotate_half(k) * s)
        kc, vc = kv
        kc[:B, :, past_len:past_len + S] = k
        vc[:B, :, past_len:past_len + S] = v
        L = past_len + S
        if self.attn_impl == "hip" and S > 1 and q.dtype == torch.float16 and D in (64, 128) and kc.is_contiguous() and vc.is_contiguous():
            # the build's own fused causal attention (onebit_attention_prefill: flash style on MFMA, csrc/ob_flash.h),
            # any past_len; q goes in token-major, the output comes back as the [B, S, H * D] rows o_proj consumes.
            # (before the GQA expansion below: the kernel indexes the kv heads itself)
            o = hip_attention_prefill(q.transpose(1, 2).contiguous(), kc, vc, past_len)
            return o_proj(o.view(B, S, H * D))
        keys, vals = kc[:B, :, :L], vc[:B, :, :L]
        if Hkv != H:
            rep = H // Hkv
            keys = keys.repeat_interleave(rep, dim=1)
            vals = vals.repeat_interleave(rep, dim=1)
        if self.attn_impl == "sdpa" and S > 1 and past_len == 0:
            # fused causal attention (the reference offers the same switch: LlamaFlashAttention2 under
            # config._flash_attn_2_enabled, modeling_bitllama.py:588,862): no [S, S] score tensor in HBM.
            # Probabilities are not rounded to fp16 on the way, so results differ from the eager path
            # within fp16 tolerance; parity tests run the eager path.
            o = nn.functional.scaled_dot_product_attention(q, keys, vals, is_causal=True)
            return o_proj(o.transpose(1, 2).contiguous().reshape(B, S, H * D))
        w = torch.matmul(q, keys.transpose(2, 3)) / math.sqrt(D)       # :546
        if S > 1:
            mask = torch.full((S, L), torch.finfo(w.dtype).min, device=w.device, dtype=w.dtype)
            mask = torch.triu(mask, diagonal=past_len + 1)
            w = w + mask[None, None]
        w = nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)   # :562
        o = torch.matmul(w, vals).transpose(1, 2).contiguous().reshape(B, S, H * D)
        return o_proj(o)


class LlamaDecoderLayerInf(nn.Module):
    """modeling_bitllama.py:856-928."""

    def __init__(self, config: OneBitLlamaConfig, dtype=None):
        super().__init__()
        self.self_attn = LlamaAttentionInf(config, dtype)
        self.mlp = LlamaMLPInf(config, dtype)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, dtype=dtype)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, dtype=dtype)

    def forward(self, h, cos, sin, kv, past_len):
        h = h + self.self_attn(self.input_layernorm(h), cos, sin, kv, past_len)
        h = h + self.mlp(self.post_attention_layernorm(h))
        return h


class LlamaModelInf(nn.Module):
    """modeling_bitllama.py:1189-1335."""

    def __init__(self, config: OneBitLlamaConfig, dtype=None):
        super().__init__()
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, dtype=dtype)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList([LlamaDecoderLayerInf(config, dtype) for _ in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps, dtype=dtype)


class KVCache:
    """Preallocated [B, Hkv, max_len, D] key/value buffers per layer."""

    def __init__(self, config: OneBitLlamaConfig, batch: int, max_len: int, device, dtype):
        shape = (batch, config.num_key_value_heads, max_len, config.head_dim)
        self.layers: List[Tuple[torch.Tensor, torch.Tensor]] = [
            (torch.zeros(shape, device=device, dtype=dtype), torch.zeros(shape, device=device, dtype=dtype))
            for _ in range(config.num_hidden_layers)]
        self.length = 0
        self.max_len = max_len


class OneBitLlamaForCausalLM(nn.Module):
    """``BitLlamaForCausalLMInf`` (modeling_bitllama.py:1512): fp16 lm_head, logits returned
    as fp32 (:1610-1611)."""

    def __init__(self, config: OneBitLlamaConfig, dtype=torch.float16):
        super().__init__()
        self.config = config
        self.model = LlamaModelInf(config, dtype)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False, dtype=dtype)
        self.lm_head.weight.requires_grad_(False)
        self._rope = None

    def _rope_tables(self, device, dtype, seq_len: Optional[int] = None):
        """The rotary caches, grown on demand as the reference's ``LlamaRotaryEmbedding.forward`` does (:106-108): built for
        max_position_embeddings, rebuilt for ``seq_len`` positions the first time a pass needs more (with "dynamic"
        scaling that rebuild also rescales the base for that length -- and, as in the reference, it stays rescaled)."""
        cfg = self.config
        have = 0 if self._rope is None else self._rope[0].shape[0]
        need = max(cfg.max_position_embeddings, seq_len or 0, have)
        if self._rope is None or self._rope[0].device != device or self._rope[0].dtype != dtype or need > have:
            self._rope = rope_tables(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta, device, dtype,
                                     seq_len=need, scaling=cfg.rope_scaling)
        return self._rope

    def set_attention(self, impl: str) -> "OneBitLlamaForCausalLM":
        """"eager" (reference op order, default), "hip" (this build's fused causal prefill attention kernel,
        ``onebit_attention_prefill``) or "sdpa" (torch's fused attention -- an AOTriton kernel on ROCm; kept for A/B)."""
        if impl not in ("eager", "sdpa", "hip"):
            raise ValueError("attention implementation must be 'eager', 'hip' or 'sdpa'")
        for layer in self.model.layers:
            layer.self_attn.attn_impl = impl
        return self

    def new_cache(self, batch: int = 1, max_len: Optional[int] = None) -> KVCache:
        p = self.lm_head.weight
        return KVCache(self.config, batch, max_len or self.config.max_position_embeddings, p.device, p.dtype)

    def set_fused_glue(self, on: bool = True) -> "OneBitLlamaForCausalLM":
        """Route the row-wise glue between the 1-bit GEMMs through ``onebit_rows_res_ln_rms`` /
        ``onebit_rows_swiglu`` (o_proj, gate, up and down_proj leave their pre-LayerNorm output; one
        fused pass does LayerNorm + residual + RMSNorm, another LayerNorm x2 + SiLU * up): fewer passes
        over the [tokens, hidden] tensors in prefill.  fp16 models, no projection bias.  The fused
        kernels use the decode engine's one-instruction LayerNorm form, so logits agree with the
        default path within fp16 tolerance, not bit for bit."""
        self.fused_glue = bool(on)
        return self

    def _forward_fused(self, input_ids, cache, past):
        from . import _lib
        from .bitnet import _stream_ptr
        lib = _lib.load()
        cfg, m = self.config, self.model
        B, S = input_ids.shape
        T, H, I = B * S, cfg.hidden_size, cfg.intermediate_size
        h = m.embed_tokens(input_ids).reshape(T, H)
        cos, sin = self._rope_tables(h.device, h.dtype, past + S)
        sp = _stream_ptr(h.device)

        def res_ln_rms(hres, u, w, consumers=(), bias_prev=None):
            """hres + LayerNorm(u) (+ bias_prev: the bias of the projection that produced u) -> new residual; RMSNorm -> x.
            ``consumers``: projections that read x; when all of them take pre-scaled rows at this T the kernel writes
            fp16(x * h_i) for each instead of x and they run with ONEBIT_FLAG_PRESCALED (no separate scaling pass).
            Returns (residual, x or None, [a_i] or None)."""
            hout = torch.empty_like(hres)
            pres = bool(consumers) and all(p.prescaled_ok(T, hres.dtype, bias_deferred=True) for p in consumers)
            x = None if pres else torch.empty_like(hres)
            xs = [torch.empty_like(hres) for _ in consumers] if pres else []
            hp = (ctypes.c_void_p * 3)(*[p.input_factor.data_ptr() for p in consumers][:len(xs)])
            xp = (ctypes.c_void_p * 3)(*[a.data_ptr() for a in xs])
            with torch.cuda.device(h.device):
                _lib.check(lib.onebit_rows_res_ln_rms_bias(hres.data_ptr(), u.data_ptr(), None if bias_prev is None else bias_prev.data_ptr(),
                                                           w.data_ptr(), hout.data_ptr(), None if x is None else x.data_ptr(), hp, xp, len(xs),
                                                           T, H, cfg.rms_norm_eps, 1e-5, sp), "onebit_rows_res_ln_rms")
            return hout, x, (xs if pres else None)

        def proj(p, x, a):
            # (a projection's bias -- q / k / v of a checkpoint with config.attention_bias -- joins its LayerNorm in the rope kernel)
            return p.pre_layernorm_bias_deferred(a, prescaled=True) if a is not None else p.pre_layernorm_bias_deferred(x)

        def proj_group(ps, x, xs_):
            """The pre-LayerNorm outputs of projections that share their input: ONE grouped launch on the producer-scaled rows
            (onebit_linear_group_prescaled) when the group is eligible, else one call each."""
            if xs_ is not None and len(ps) > 1:
                from .engine import _Proj, _proj
                try:
                    arr = (_Proj * len(ps))(*[_proj(p, allow_bias=True) for p in ps])
                except ValueError:                      # shapes the fused structs do not describe: one call per projection
                    arr = None
                if arr is not None:
                    us = [torch.empty((T, p.out_features), dtype=h.dtype, device=h.device) for p in ps]
                    up = (ctypes.c_void_p * 3)(*[u_.data_ptr() for u_ in us])
                    ap = (ctypes.c_void_p * 3)(*[a_.data_ptr() for a_ in xs_])
                    with torch.cuda.device(h.device):
                        rc = lib.onebit_linear_group_prescaled(ctypes.cast(arr, ctypes.c_void_p), up, ap, len(ps), T, sp)
                    if rc == 0:
                        return us
            return [proj(p, x, None if xs_ is None else a_) for p, a_ in zip(ps, xs_ if xs_ is not None else [None] * len(ps))]

        row_arrays = None          # (slot, position) of every token row: the ragged rope kernel, which takes the q / k / v biases

        x, xs = m.layers[0].input_layernorm(h), None
        u_down = None
        for li, (layer, kv) in enumerate(zip(m.layers, cache.layers)):
            att = layer.self_attn
            qkv_bias = att.q_proj.bias is not None
            if qkv_bias != (att.k_proj.bias is not None) or qkv_bias != (att.v_proj.bias is not None):
                raise ValueError("fused glue: a bias on all of q / k / v_proj or on none (config.attention_bias)")
            fused_attn = (att.attn_impl in ("sdpa", "hip") and S > 1 and (past == 0 or att.attn_impl == "hip")
                          and (att.attn_impl == "sdpa" or att.head_dim in (64, 128))
                          and kv[0].is_contiguous() and kv[1].is_contiguous()
                          and att.head_dim >= 16 and att.head_dim & (att.head_dim - 1) == 0
                          # onebit_rows_qkv_rope writes rows [b < B][kv head][past + s][D] through raw pointers: the
                          # cache must really have B slots of that geometry in the activations' dtype (a batch-1
                          # cache reused with B > 1 takes the checked torch path, which raises on the shape)
                          and all(c.dim() == 4 and c.shape[0] >= B and c.dtype == h.dtype and c.device == h.device
                                  and tuple(c.shape[1:]) == (att.num_key_value_heads, kv[0].shape[2], att.head_dim)
                                  for c in kv)
                          and past + S <= kv[0].shape[2])
            if u_down is not None:
                h, x, xs = res_ln_rms(h, u_down, layer.input_layernorm.weight,
                                      (att.q_proj, att.k_proj, att.v_proj) if fused_attn else ())
            if fused_attn:
                # q|k|v LayerNorm + RoPE + head transpose in one pass (onebit_rows_qkv_rope): k, v land in
                # the cache rows, q in token-major [B, S, heads, D]; then the fused causal attention
                Hh, Hkv, D = att.num_heads, att.num_key_value_heads, att.head_dim
                u_q, u_k, u_v = proj_group((att.q_proj, att.k_proj, att.v_proj), x, xs)
                q = torch.empty((B, S, Hh, D), dtype=h.dtype, device=h.device)     # token-major: sdpa returns the same layout
                kc, vc = kv
                with torch.cuda.device(h.device):
                    if qkv_bias:
                        # biases join the LayerNorm inside the ragged form of the kernel: row t = (b, s) -> slot b, position past + s
                        if row_arrays is None:
                            ar = torch.arange(T, device=h.device, dtype=torch.int32)
                            row_arrays = ((ar // S).contiguous(), (past + ar % S).contiguous())
                        bq, bk, bv = (p_.bias.to(h.dtype).contiguous() for p_ in (att.q_proj, att.k_proj, att.v_proj))
                        _lib.check(lib.onebit_rows_qkv_rope_ragged(u_q.data_ptr(), u_k.data_ptr(), u_v.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                                   row_arrays[0].data_ptr(), row_arrays[1].data_ptr(), q.data_ptr(), kc.data_ptr(),
                                                                   vc.data_ptr(), bq.data_ptr(), bk.data_ptr(), bv.data_ptr(), T, Hh, Hkv, D,
                                                                   kc.shape[0], kc.shape[2], cos.shape[0], 1e-5, sp), "onebit_rows_qkv_rope_ragged")
                    else:
                        _lib.check(lib.onebit_rows_qkv_rope(u_q.data_ptr(), u_k.data_ptr(), u_v.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                            q.data_ptr(), kc.data_ptr(), vc.data_ptr(), B, S, Hh, Hkv, D, past,
                                                            kc.shape[2], cos.shape[0], 1e-5, _lib.FLAG_Q_TOKEN_MAJOR, sp), "onebit_rows_qkv_rope")
                if att.attn_impl == "hip":
                    # own flash kernel: rows come back token-major, already multiplied by o_proj's input_factor when o_proj
                    # takes pre-scaled rows at this T (no separate scaling pass left on the route)
                    o_pres = att.o_proj.prescaled_ok(T, h.dtype, bias_deferred=True)
                    o = hip_attention_prefill(q, kc, vc, past, att.o_proj.input_factor if o_pres else None).view(T, Hh * D)
                    u_o = att.o_proj.pre_layernorm_bias_deferred(o, prescaled=o_pres)
                else:
                    keys, vals = kc[:B, :, :S], vc[:B, :, :S]
                    if Hkv != Hh:
                        keys, vals = keys.repeat_interleave(Hh // Hkv, dim=1), vals.repeat_interleave(Hh // Hkv, dim=1)
                    o = nn.functional.scaled_dot_product_attention(q.transpose(1, 2), keys, vals, is_causal=True)
                    u_o = att.o_proj.pre_layernorm_bias_deferred(o.transpose(1, 2).contiguous().reshape(T, Hh * D))    # (no copy when o is token-major)
            else:
                u_o = att.forward(x.view(B, S, H), cos, sin, kv, past, pre_ln_out=True).reshape(T, H)
            mlp = layer.mlp
            h, x, xs = res_ln_rms(h, u_o, layer.post_attention_layernorm.weight, (mlp.gate_proj, mlp.up_proj), bias_prev=att.o_proj.bias)
            u_g, u_u = proj_group((mlp.gate_proj, mlp.up_proj), x, xs)
            act = torch.empty_like(u_g)
            down_pres = mlp.down_proj.prescaled_ok(T, h.dtype)
            with torch.cuda.device(h.device):
                _lib.check(lib.onebit_rows_swiglu(u_g.data_ptr(), u_u.data_ptr(),
                                                  mlp.down_proj.input_factor.data_ptr() if down_pres else None,
                                                  act.data_ptr(), T, I, 1e-5, sp), "onebit_rows_swiglu")
            u_down = mlp.down_proj.pre_layernorm_prescaled(act) if down_pres else mlp.down_proj.pre_layernorm(act)
        h, x, _ = res_ln_rms(h, u_down, m.norm.weight)
        return self.lm_head(x.view(B, S, H)).float()

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, cache: Optional[KVCache] = None) -> torch.Tensor:
        """input_ids [B, S] -> fp32 logits [B, S, vocab]; appends to ``cache`` when given."""
        B, S = input_ids.shape
        if cache is None:
            cache = self.new_cache(B, S)
        past = cache.length
        if past + S > cache.max_len:
            raise ValueError("KV cache too small")
        if getattr(self, "fused_glue", False) and self.lm_head.weight.dtype == torch.float16 and B * S > 1:
            logits = self._forward_fused(input_ids, cache, past)
            cache.length = past + S
            return logits
        h = self.model.embed_tokens(input_ids)
        cos, sin = self._rope_tables(h.device, h.dtype, past + S)
        for layer, kv in zip(self.model.layers, cache.layers):
            h = layer(h, cos, sin, kv, past)
        cache.length = past + S
        h = self.model.norm(h)
        return self.lm_head(h).float()

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, max_new_tokens: int, eos_token_id=None, pad_token_id=None) -> torch.Tensor:
        """Greedy search (generation/utils.py:2338, loop :2491-2540): argmax of the last position.
        With ``eos_token_id`` (int or list) a row that produced EOS is finished: its later tokens are
        ``pad_token_id`` (required then for B > 1, as in the reference, :2543-2546) and generation stops
        when every row is finished (:2562-2570).  Rows of ``input_ids`` must have equal length: there is
        no attention_mask / left padding on this path (the reference's padded batches are not reproduced)."""
        B, S = input_ids.shape
        cache = self.new_cache(B, S + max_new_tokens)
        eos = None
        if eos_token_id is not None:
            eos = torch.tensor(list(eos_token_id) if isinstance(eos_token_id, (list, tuple, set)) else [int(eos_token_id)],
                               device=input_ids.device)
            if pad_token_id is None:
                if B > 1:
                    raise ValueError("generate: eos_token_id with batch > 1 needs pad_token_id (generation/utils.py:2543)")
                pad_token_id = int(eos[0])
        unfinished = torch.ones(B, 1, dtype=torch.long, device=input_ids.device)
        logits = self.forward(input_ids, cache)
        out = [input_ids]
        for i in range(max_new_tokens):
            nxt = logits[:, -1].argmax(-1, keepdim=True)
            if eos is not None:
                nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
            out.append(nxt)
            if eos is not None:
                unfinished = unfinished * (nxt != eos.view(1, -1)).all(dim=1, keepdim=True).long()
                if int(unfinished.max()) == 0:
                    break
            if i + 1 < max_new_tokens:
                logits = self.forward(nxt, cache)
        return torch.cat(out, dim=1)


def _generate_native(self, input_ids: torch.Tensor, max_new_tokens: int, eos_token_id=None, pad_token_id=None) -> torch.Tensor:
    """``generate`` on the native engines, same return contract: B = 1 -> ``DecodeEngine`` (the prompt through ``onebit_mixed_step``,
    every further token one HIP-graph replay of ``onebit_decode_step``); B > 1 -> ``ContinuousBatcher`` with one slot per row (the
    prompts enter in ONE mixed step, decode-only steps replay the batched step's graph).  The steps are enqueued without looking at
    the tokens; EOS is applied afterwards exactly as ``generate`` applies it (a finished row continues with ``pad_token_id``, the
    output ends where every row is finished: generation/utils.py:2543-2570).  Engines are kept per (batch, max_len) on the model.
    Greedy tokens equal ``generate``'s up to fp16 near-ties of the logits (the engines' kernels sum in another order)."""
    from .engine import DecodeEngine
    from .serving import ContinuousBatcher
    if input_ids.dim() != 2:
        raise ValueError("input_ids must be [B, S]")
    B, S = input_ids.shape
    if max_new_tokens < 1:
        return input_ids
    eos = None
    if eos_token_id is not None:
        eos = set(eos_token_id) if isinstance(eos_token_id, (list, tuple, set)) else {int(eos_token_id)}
        if pad_token_id is None:
            if B > 1:
                raise ValueError("generate: eos_token_id with batch > 1 needs pad_token_id (generation/utils.py:2543)")
            pad_token_id = next(iter(eos)) if not isinstance(eos_token_id, (list, tuple)) else int(eos_token_id[0])
    dev = self.lm_head.weight.device
    engines = self.__dict__.setdefault("_native_engines", {})
    max_len = S + max_new_tokens
    key = (B, -(-max_len // 256) * 256)
    if key[1] > self.config.max_position_embeddings:
        key = (B, self.config.max_position_embeddings)
    if max_len > key[1]:
        raise ValueError("prompt + max_new_tokens exceeds max_position_embeddings")
    if B == 1:
        eng = engines.get(key)
        if eng is None:
            if len(engines) >= 4:
                engines.clear()
            eng = engines[key] = DecodeEngine(self, max_len=key[1])
        rows = [eng.generate(input_ids, max_new_tokens)[0, S:].tolist()]
    else:
        cb = engines.get(key)
        if cb is None:
            if len(engines) >= 4:
                engines.clear()
            cb = engines[key] = ContinuousBatcher(self, max_batch=B, max_len=key[1])
        ids = [cb.add_request(r, max_new_tokens) for r in input_ids.tolist()]
        out = cb.run()
        rows = [out[i] for i in ids]
        for i in ids:
            cb.sched.finished.pop(i, None)
    if eos is not None:
        ends = [next((j + 1 for j, t in enumerate(r) if t in eos), len(r)) for r in rows]
        n = max(ends)
        rows = [r[:e] + [pad_token_id] * (n - e) for r, e in zip(rows, ends)]
    new = torch.tensor(rows, dtype=input_ids.dtype, device=dev)
    return torch.cat([input_ids.to(dev), new], dim=1)


OneBitLlamaForCausalLM.generate_native = torch.no_grad()(_generate_native)


def synthetic_state_dict(config: OneBitLlamaConfig, seed: int = 0, dtype=torch.float16, device="cpu"):
    """Seeded synthetic OneBit checkpoint in the reference's on-disk key layout (SURVEY.md 3.4 /
    8d): packed W = uniform random bytes, h, g = 0.1*U(0.5,1.5) with 10% sign flips, RMSNorm = 1,
    embeddings N(0,1), lm_head N(0, 0.02).  ``device="cpu"`` is bit-reproducible (fixtures);
    a GPU device generates the same distribution quickly for full-size benchmarks."""
    sd = {}
    device = torch.device(device)
    H, I, D = config.hidden_size, config.intermediate_size, config.head_dim
    Hq, Hkv = config.num_attention_heads * D, config.num_key_value_heads * D
    projs = [("self_attn.q_proj", H, Hq), ("self_attn.k_proj", H, Hkv), ("self_attn.v_proj", H, Hkv),
             ("self_attn.o_proj", Hq, H), ("mlp.gate_proj", H, I), ("mlp.up_proj", H, I),
             ("mlp.down_proj", I, H)]

    def scale(n, g):
        v = 0.1 * (0.5 + torch.rand(n, generator=g, device=device))
        flip = torch.where(torch.rand(n, generator=g, device=device) < 0.1, -1.0, 1.0)
        return (v * flip).to(dtype)

    for l in range(config.num_hidden_layers):
        for p, (name, K, N) in enumerate(projs):
            g = torch.Generator(device=device).manual_seed(seed + 1000 * l + p)
            pre = f"model.layers.{l}.{name}."
            sd[pre + "weight"] = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8,
                                               device=device).view(torch.int8)
            sd[pre + "input_factor"] = scale(K, g)
            sd[pre + "weight_scale"] = scale(N, g)
            if config.attention_bias and name.startswith("self_attn."):      # bias=config.attention_bias, modeling_bitllama.py:451-454
                gb = torch.Generator(device=device).manual_seed(seed + 1000 * l + p + 500)     # own stream: other draws unchanged
                sd[pre + "bias"] = (0.1 * torch.randn(N, generator=gb, device=device)).to(dtype)
        sd[f"model.layers.{l}.input_layernorm.weight"] = torch.ones(H, dtype=dtype, device=device)
        sd[f"model.layers.{l}.post_attention_layernorm.weight"] = torch.ones(H, dtype=dtype, device=device)
    g = torch.Generator(device=device).manual_seed(seed + 999_983)
    sd["model.embed_tokens.weight"] = torch.randn(config.vocab_size, H, generator=g, device=device).to(dtype)
    sd["model.norm.weight"] = torch.ones(H, dtype=dtype, device=device)
    sd["lm_head.weight"] = (0.02 * torch.randn(config.vocab_size, H, generator=g, device=device)).to(dtype)
    return sd


def build_synthetic_model(config: OneBitLlamaConfig, seed: int = 0, dtype=torch.float16,
                          device="cuda") -> "OneBitLlamaForCausalLM":
    """Random-init model of the given architecture with weights generated on `device`."""
    with torch.device("meta"):
        model = OneBitLlamaForCausalLM(config, dtype)
    model = model.to_empty(device=device)
    sd = synthetic_state_dict(config, seed, dtype, device)
    model.load_state_dict(sd, assign=True)
    for p in model.parameters():
        p.requires_grad_(False)
    return model.eval()
