"""Fused whole-token greedy decode (batch 1) -- SURVEY.md section 8(f) rank 1.

``DecodeEngine`` drives ``onebit_decode_step`` (include/onebit.h): 5 HIP launches per decoder
layer + lm_head + argmax, every 1-bit projection read once in the reference's packed layout,
LayerNorm / RMSNorm / RoPE / SiLU / residuals fused into the consumers' prologues, token id and
position kept on the device.  One decode step is captured into a HIP graph (via
``torch.cuda.CUDAGraph``) and replayed per token -- 160+ launches per token would otherwise be
host-bound (the reference issues ~10 ATen launches per BitLinearInf call).

Prefill of the prompt (S > 1) runs through the module path (``OneBitLlamaForCausalLM.forward``)
into the same preallocated KV cache.
"""
from __future__ import annotations

import copy
import ctypes
from typing import List

import torch

from . import _lib
from .bitnet import BitLinearInf
from .llama import KVCache, OneBitLlamaForCausalLM

_vp, _i64, _i32, _f = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float


class _Proj(ctypes.Structure):
    _fields_ = [("weight", _vp), ("input_factor", _vp), ("weight_scale", _vp),
                ("N", _i64), ("K", _i64), ("ldw_bytes", _i64)]


class _Layer(ctypes.Structure):
    _fields_ = [("q", _Proj), ("k", _Proj), ("v", _Proj), ("o", _Proj), ("gate", _Proj), ("up", _Proj),
                ("down", _Proj), ("input_layernorm_w", _vp), ("post_attention_layernorm_w", _vp),
                ("k_cache", _vp), ("v_cache", _vp),
                ("q_bias", _vp), ("k_bias", _vp), ("v_bias", _vp), ("o_bias", _vp)]          # ABI 8: config.attention_bias


class _Model(ctypes.Structure):
    _fields_ = [("n_layers", _i32), ("hidden", _i32), ("intermediate", _i32), ("n_heads", _i32),
                ("n_kv_heads", _i32), ("head_dim", _i32), ("vocab", _i32), ("max_len", _i32),
                ("rms_eps", _f), ("ln_eps", _f), ("layers", ctypes.POINTER(_Layer)),
                ("embed", _vp), ("final_norm_w", _vp), ("lm_head", _vp), ("rope_cos", _vp), ("rope_sin", _vp)]


class _State(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint64),      # ABI 8: checked by the library (a state of another ABI is refused, not misread)
                ("token", _vp), ("pos", _vp), ("out_tokens", _vp), ("max_out", _i32),
                ("hres0", _vp), ("hres1", _vp), ("u_q", _vp), ("u_k", _vp), ("u_v", _vp),
                ("attn_out", _vp), ("u_o", _vp), ("u_gate", _vp), ("u_up", _vp), ("u_down", _vp),
                ("logits", _vp), ("part_val", _vp), ("part_idx", _vp), ("attn_splits", _i32), ("attn_scratch", _vp),
                ("tile_stats", _vp), ("rope_cur", _vp), ("attn_blind", _i32),
                ("attn_chunk", _i32), ("q_rows", _vp)]                                   # ABI 9: key-block long-context attention


class _BatchState(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint64), ("batch", _i32), ("tokens", _vp), ("pos", _vp), ("hres0", _vp), ("hres1", _vp), ("x", _vp),
                ("act", _vp), ("u_q", _vp), ("u_k", _vp), ("u_v", _vp), ("attn_out", _vp), ("u_o", _vp),
                ("u_gate", _vp), ("u_up", _vp), ("u_down", _vp),
                ("next_tokens", _vp), ("logits", _vp), ("part_val", _vp), ("part_idx", _vp), ("qkv_stats", _vp),
                ("x_scaled", _vp), ("chains", _i32),
                ("attn_splits", _i32), ("attn_chunk", _i32), ("q_rows", _vp), ("attn_scratch", _vp)]      # ABI 9: key-block attention


class _KState(ctypes.Structure):      # onebit_kshard_state_t (ABI 8; the attention fields: ABI 9)
    _fields_ = [("struct_size", ctypes.c_uint64), ("token", _vp), ("pos", _vp), ("out_tokens", _vp), ("max_out", _i32),
                ("hres0", _vp), ("hres1", _vp), ("x", _vp), ("u_q", _vp), ("u_k", _vp), ("u_v", _vp), ("attn_out", _vp),
                ("u_gate", _vp), ("u_up", _vp), ("act", _vp), ("u_down", _vp),
                ("z_qkv", _vp), ("z_o", _vp), ("z_gu", _vp), ("z_down", _vp),
                ("logits", _vp), ("part_val", _vp), ("part_idx", _vp), ("tile_stats", _vp),
                ("k0_hidden", _i32), ("k0_attn", _i32), ("k0_inter", _i32),
                ("attn_chunk", _i32), ("attn_splits", _i32), ("attn_scratch", _vp)]


KSEG_QKV, KSEG_ATTN_O, KSEG_GATE_UP, KSEG_DOWN, KSEG_HEAD = 0, 1, 2, 3, 4


class _Seg(ctypes.Structure):         # onebit_seg_t (ABI 9)
    _fields_ = [("row0", _i32), ("n", _i32), ("slot", _i32), ("past", _i32)]


class _MixedState(ctypes.Structure):  # onebit_mixed_state_t (ABI 9)
    _fields_ = [("struct_size", ctypes.c_uint64), ("n_rows", _i32), ("n_dec", _i32), ("n_seg", _i32), ("n_out", _i32),
                ("n_slots", _i32), ("attn_chunk", _i32), ("dec_ctx", _i32),
                ("tokens", _vp), ("row_slot", _vp), ("row_pos", _vp), ("segs", ctypes.POINTER(_Seg)), ("out_rows", _vp),
                ("next_tokens", _vp), ("logits", _vp), ("part_val", _vp), ("part_idx", _vp),
                ("workspace", _vp), ("workspace_bytes", ctypes.c_size_t)]


class _FusedIn(ctypes.Structure):
    _fields_ = [("xin", _vp), ("embed", _vp), ("hres_in", _vp), ("u_prev", _vp), ("rms_w", _vp),
                ("u_gate", _vp), ("u_up", _vp), ("token", _vp), ("hres_out", _vp),
                ("rms_eps", _f), ("ln_eps", _f), ("st_prev", _vp), ("st_gate", _vp), ("st_up", _vp),
                ("st_out", _vp * 3)]


PRO_PLAIN, PRO_EMBED_RMS, PRO_RES_LN_RMS, PRO_SWIGLU = 0, 1, 2, 3


def tile_stats_floats(n: int) -> int:
    """fp32 elements of the per-tile LayerNorm partials of an n-vector (whole blocks of 256 tiles)."""
    return ((n + 4095) // 4096) * 512


def fused_gemv(mods, outs, prologue, rms_eps=1e-6, ln_eps=1e-5, stats_out=None, **inputs):
    """One ``onebit_fused_gemv`` launch on the current stream: projections ``mods`` (BitLinearInf
    sharing in_features) write their pre-LayerNorm outputs to ``outs`` (fp16 [N_i] tensors);
    ``inputs`` are the prologue tensors named as in ``onebit_fused_in_t`` (``st_prev`` / ``st_gate``
    / ``st_up``: tile partials of the corresponding vectors, fp32 ``tile_stats_floats(n)``);
    ``stats_out[i]``: where projection i publishes its own partials (or None)."""
    lib = _lib.load()
    n = len(mods)
    projs = (_Proj * n)(*[_proj(m) for m in mods])
    optr = (_vp * n)(*[o.data_ptr() for o in outs])
    ptr = lambda k: inputs[k].data_ptr() if inputs.get(k) is not None else None
    so = list(stats_out or []) + [None] * 3
    st_out = (_vp * 3)(*[t.data_ptr() if t is not None else None for t in so[:3]])
    fin = _FusedIn(ptr("xin"), ptr("embed"), ptr("hres_in"), ptr("u_prev"), ptr("rms_w"), ptr("u_gate"),
                   ptr("u_up"), ptr("token"), ptr("hres_out"), rms_eps, ln_eps, ptr("st_prev"), ptr("st_gate"),
                   ptr("st_up"), st_out)
    dev = outs[0].device
    with torch.cuda.device(dev):
        rc = lib.onebit_fused_gemv(ctypes.cast(projs, _vp), ctypes.cast(optr, _vp), n, prologue,
                                   ctypes.cast(ctypes.pointer(fin), _vp),
                                   torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "onebit_fused_gemv")


def fp16_view(model: OneBitLlamaForCausalLM) -> OneBitLlamaForCausalLM:
    """The model the fused engines run: every floating parameter in fp16, the packed int8 weights SHARED with
    ``model`` (not copied).  An fp16 model is returned as is.  This is what the reference's
    ``from_pretrained(..., torch_dtype=torch.float16)`` does to the released FP32 checkpoints
    (/root/reference/checkpoints/README.md:10; modeling_utils.py:696 casts floating parameters only), done once
    at engine build instead of making the caller find out; the caller's fp32 model is left untouched."""
    if all(p.dtype == torch.float16 for p in model.parameters() if p.is_floating_point()):
        return model
    memo = {id(p): p for p in model.parameters() if not p.is_floating_point()}
    return copy.deepcopy(model, memo).half()


def _proj(m: BitLinearInf, allow_bias: bool = False, kslice=None) -> _Proj:
    if m.bias is not None and not allow_bias:
        raise ValueError(
            "the fused decode engines take a projection bias on q / k / v / o_proj only (config.attention_bias, "
            "modeling_bitllama.py:451-454: the reference's MLP projections have none); decode this checkpoint through the module "
            "path instead: model.generate(...) or ContinuousBatcher(..., native=False)")
    if m.bias is not None and (m.bias.dtype != torch.float16 or not m.bias.is_contiguous()):
        raise ValueError("the fused decode engines run fp16 parameters: pass the model through engine.fp16_view()")
    if m.weight_scale.dtype != torch.float16 or m.input_factor.dtype != torch.float16:
        raise ValueError("the fused decode engines run fp16 parameters: pass the model through engine.fp16_view()")
    if m.in_features % 32 != 0:
        raise ValueError(f"DecodeEngine: in_features={m.in_features} is not a multiple of 32; use the module path")
    w = m.weight
    if w.stride(1) != 1 or not m.weight_scale.is_contiguous() or not m.input_factor.is_contiguous():
        raise ValueError("DecodeEngine: parameters must be contiguous")
    if kslice is not None:
        # the rank's K slice IN PLACE: a byte-column window of the packed matrix (the ABI takes the row pitch), h[k0:k1]
        k0, k1 = kslice
        return _Proj(w.data_ptr() + k0 // 8, m.input_factor.data_ptr() + 2 * k0, m.weight_scale.data_ptr(),
                     m.out_features, k1 - k0, w.stride(0))
    return _Proj(w.data_ptr(), m.input_factor.data_ptr(), m.weight_scale.data_ptr(),
                 m.out_features, m.in_features, w.stride(0))


def _model_struct(model: OneBitLlamaForCausalLM, caches, max_len: int, krange=None):
    """ctypes image of ``onebit_model_t`` for ``model`` with per-layer (k, v) cache tensors; returns
    (struct, objects that must stay alive as long as the struct is used).  ``krange(K) -> (k0, k1)``: describe every
    projection's K SLICE instead (onebit_decode_step_ksharded: the rank's window into the full packed matrices)."""
    cfg = model.config
    p = model.lm_head.weight
    dev, f16 = p.device, torch.float16
    cos, sin = model._rope_tables(dev, f16)
    cos, sin = cos.contiguous(), sin.contiguous()
    layers = (_Layer * cfg.num_hidden_layers)()
    for i, (layer, (kc, vc)) in enumerate(zip(model.model.layers, caches)):
        a, mlp = layer.self_attn, layer.mlp
        for w in (layer.input_layernorm.weight, layer.post_attention_layernorm.weight):
            if w.dtype != f16:
                raise ValueError("the decode steps need fp16 RMSNorm weights")
        qkv_b = [p_.bias is not None for p_ in (a.q_proj, a.k_proj, a.v_proj)]
        if any(qkv_b) and not all(qkv_b):
            raise ValueError("the fused decode engines need a bias on all of q / k / v_proj or on none (config.attention_bias)")
        bptr = lambda p_: None if p_.bias is None else p_.bias.data_ptr()
        ks = (lambda p_: None) if krange is None else (lambda p_: krange(p_.in_features))
        layers[i] = _Layer(_proj(a.q_proj, True, ks(a.q_proj)), _proj(a.k_proj, True, ks(a.k_proj)), _proj(a.v_proj, True, ks(a.v_proj)),
                           _proj(a.o_proj, True, ks(a.o_proj)),
                           _proj(mlp.gate_proj, False, ks(mlp.gate_proj)), _proj(mlp.up_proj, False, ks(mlp.up_proj)),
                           _proj(mlp.down_proj, False, ks(mlp.down_proj)),
                           layer.input_layernorm.weight.data_ptr(),
                           layer.post_attention_layernorm.weight.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                           bptr(a.q_proj), bptr(a.k_proj), bptr(a.v_proj), bptr(a.o_proj))
    m = _Model(cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads,
               cfg.num_key_value_heads, cfg.head_dim, cfg.vocab_size, max_len, cfg.rms_norm_eps, 1e-5, layers,
               model.model.embed_tokens.weight.data_ptr(), model.model.norm.weight.data_ptr(),
               model.lm_head.weight.data_ptr(), cos.data_ptr(), sin.data_ptr())
    return m, [layers, cos, sin]


class DecodeEngine:
    def __init__(self, model: OneBitLlamaForCausalLM, max_len: int, use_graph: bool = True,
                 long_context_from: int = 160, attn_splits: int = 8, long_attention: str = "keyblock", attn_chunk: int = 128,
                 native_prefill: bool = True, prefill_rows: int = 2048):
        """``native_prefill``: ``generate`` / ``prime`` run the prompt through ``onebit_mixed_step`` (one C call per ``prefill_rows``
        prompt tokens, lm_head on the last row only) instead of the module path -- time to first token on 7B shapes 16.5 -> 2.3 ms at 16
        prompt tokens, 19.7 -> 5.3 at 128, 23.5 -> 12.2 at 512, 66 -> 24.5 at 2040 (tools/ttft_probe.py).  ``prefill`` keeps the module
        path: it returns the logits of every prompt row.
        ``long_context_from``: position from which a step uses a split-KV attention graph; below it one workgroup per head
        with the first 128 positions' scores in registers is faster (measured crossover on 7B: 0.947 vs 0.963 ms / token at 132-164
        cached tokens, 1.184 vs 0.971 at 260-292: tools/ctx_probe.py).  0 disables.
        ``long_attention``: "keyblock" (round 6, default: LayerNorm + RoPE + cache append in one launch, then
        ``onebit_attention_decode_rows`` -- ``attn_chunk`` positions per workgroup, K and V read once, last-arriver combine; one
        HIP graph per power-of-two split count, chosen per step by the host-known position) or "pair" (round 2: scores kernel +
        P.V kernel with ``attn_splits`` splits, the reference's fp16 probability rounding)."""
        if long_attention not in ("keyblock", "pair"):
            raise ValueError("long_attention must be 'keyblock' or 'pair'")
        cfg = model.config
        if not model.lm_head.weight.is_cuda:
            raise RuntimeError("DecodeEngine needs the model on a ROCm GPU (no CPU fallback)")
        model = fp16_view(model)        # an fp32 checkpoint: floating parameters cast once, packed weights shared
        p = model.lm_head.weight
        self.model, self.cfg, self.dev = model, cfg, p.device
        self.lib = _lib.load()
        if not hasattr(self.lib, "onebit_decode_step"):
            raise _lib.OneBitLibraryError("libonebit_hip.so lacks onebit_decode_step")
        self.max_len = int(max_len)
        if self.max_len > cfg.max_position_embeddings:
            raise ValueError("max_len exceeds max_position_embeddings")
        dev, f16 = self.dev, torch.float16
        H, I, D = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
        self.cache = KVCache(cfg, 1, self.max_len, dev, f16)
        self._model, self._keep = _model_struct(model, self.cache.layers, self.max_len)
        self._native_prefill, self._prefill_rows, self._mixed = bool(native_prefill), max(int(prefill_rows), 64), None
        z = lambda n, dt=f16: torch.zeros(n, dtype=dt, device=dev)
        self.token = z(1, torch.int32)
        self.pos = z(1, torch.int32)
        self.out_tokens = z(self.max_len, torch.int32)
        Hq, Hkv = cfg.num_attention_heads * D, cfg.num_key_value_heads * D
        self.buf = dict(hres0=z(H), hres1=z(H), u_q=z(Hq), u_k=z(Hkv), u_v=z(Hkv), attn_out=z(Hq), u_o=z(H),
                        u_gate=z(I), u_up=z(I), u_down=z(H), logits=z(cfg.vocab_size),
                        part_val=z(1024, torch.float32), part_idx=z(1024, torch.int32))
        b = self.buf
        self._state = _State(ctypes.sizeof(_State), self.token.data_ptr(), self.pos.data_ptr(), self.out_tokens.data_ptr(), self.max_len,
                             b["hres0"].data_ptr(), b["hres1"].data_ptr(), b["u_q"].data_ptr(), b["u_k"].data_ptr(),
                             b["u_v"].data_ptr(), b["attn_out"].data_ptr(), b["u_o"].data_ptr(), b["u_gate"].data_ptr(),
                             b["u_up"].data_ptr(), b["u_down"].data_ptr(), b["logits"].data_ptr(),
                             b["part_val"].data_ptr(), b["part_idx"].data_ptr(), 0, None, None, None, 0, 0, None)
        self._rope_cur = z(2 * D)
        self._state.rope_cur = self._rope_cur.data_ptr()
        self.lib.onebit_decode_stats_floats.restype = ctypes.c_size_t
        self.lib.onebit_decode_stats_floats.argtypes = [ctypes.POINTER(_Model)]
        self._tile_stats = torch.zeros(max(int(self.lib.onebit_decode_stats_floats(ctypes.byref(self._model))), 1),
                                       dtype=torch.float32, device=dev)
        self._state.tile_stats = self._tile_stats.data_ptr()
        self.lib.onebit_decode_step.restype = ctypes.c_int
        self.lib.onebit_decode_step.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_State), _vp]
        self.graph = self.graph_long = self.graph64 = None
        self._prompt_len = 0
        self._steps = 0                     # host-side count of tokens in the cache (chooses the graph)
        self._long_from = int(long_context_from) if long_context_from and self.max_len > long_context_from else 0
        # the one-workgroup-per-head attention keeps the scores of every position in LDS (64 KiB limit
        # in onebit_decode_step): beyond that only the split-KV form exists, and it serves every position
        self._short_ok = 512 + 3 * 128 * 2 + 8 * 128 * 4 + 4 * self.max_len <= 64 * 1024
        if not self._short_ok:
            if not long_context_from:
                raise ValueError(f"max_len {self.max_len} needs the split-KV attention (long_context_from > 0)")
            self._long_from = 1
        # steps at positions < 64: the attention requests half the rows before it knows the position (onebit.h, attn_blind);
        # the same arithmetic, chosen per step by the host-known position like the split-KV graph below
        self._state64 = _State.from_buffer_copy(self._state)
        self._state64.attn_blind = 64
        if self._short_ok:
            for st64 in (False, True):
                self._launch(blind64=st64)  # warm-up: validates arguments, sets function attributes
                torch.cuda.synchronize(dev)
                if use_graph:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._launch(blind64=st64)
                    if st64:
                        self.graph64 = g
                    else:
                        self.graph = g
        self._keyblock = long_attention == "keyblock" and bool(self._long_from)
        self._kb_graphs, self._kb_states = {}, {}
        if self._keyblock:
            # one state / graph per power-of-two split count: splits * attn_chunk covers the step's context + 1
            if attn_chunk < 64 or attn_chunk % 64:
                raise ValueError("attn_chunk must be a positive multiple of 64")
            self._kb_chunk = int(attn_chunk)
            top = -(-self.max_len // self._kb_chunk)
            if top > 64:
                self._kb_chunk = 64 * -(-self.max_len // (64 * 64))          # at most 64 splits: widen the chunk
                top = -(-self.max_len // self._kb_chunk)
            self._q_rows = z(Hq)
            self.lib.onebit_attention_decode_scratch_bytes.restype = ctypes.c_size_t
            ns = 1
            while True:
                nb = int(self.lib.onebit_attention_decode_scratch_bytes(1, cfg.num_attention_heads, ns))
                scratch = torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)
                stt = _State.from_buffer_copy(self._state)
                stt.attn_splits, stt.attn_chunk, stt.q_rows, stt.attn_scratch = ns, self._kb_chunk, self._q_rows.data_ptr(), scratch.data_ptr()
                self._kb_states[ns] = (stt, scratch)
                self.pos.zero_()
                self._launch(state=stt)
                torch.cuda.synchronize(dev)
                if use_graph:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._launch(state=stt)
                    self._kb_graphs[ns] = g
                if ns >= top:
                    break
                ns *= 2
        elif self._long_from:
            S = max(2, min(int(attn_splits), 16, self.max_len // 128))
            self.lib.onebit_attn_scratch_bytes.restype = ctypes.c_size_t
            self.lib.onebit_attn_scratch_bytes.argtypes = [ctypes.POINTER(_Model), ctypes.c_int]
            nbytes = self.lib.onebit_attn_scratch_bytes(ctypes.byref(self._model), S)
            self._scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            self._state_long = _State.from_buffer_copy(self._state)
            self._state_long.attn_splits = S
            self._state_long.attn_scratch = self._scratch.data_ptr()
            self.pos.zero_()
            self._launch(long=True)
            torch.cuda.synchronize(dev)
            if use_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch(long=True)
                self.graph_long = g
        self.pos.zero_()
        self.token.zero_()

    def _kb_splits(self, ctx: int) -> int:
        """Power-of-two split count whose splits * attn_chunk positions cover a context of ``ctx`` tokens."""
        ns = 1
        while ns * self._kb_chunk < ctx and ns * 2 in self._kb_states:
            ns *= 2
        return ns

    def _launch(self, long: bool = False, blind64: bool = False, state=None):
        st = state if state is not None else (self._state_long if long else (self._state64 if blind64 else self._state))
        with torch.cuda.device(self.dev):
            rc = self.lib.onebit_decode_step(ctypes.byref(self._model), ctypes.byref(st),
                                             torch.cuda.current_stream(self.dev).cuda_stream)
        _lib.check(rc, "onebit_decode_step")

    @torch.no_grad()
    def prefill(self, input_ids: torch.Tensor) -> torch.Tensor:
        """Run the prompt through the module path into the shared KV cache; returns the fp32 logits
        and arms the engine with the first greedy token."""
        if input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise ValueError("DecodeEngine is batch 1: input_ids must be [1, S]")
        S = input_ids.shape[1]
        if S + 1 > self.max_len:
            raise ValueError("prompt longer than max_len")
        self.cache.length = 0
        logits = self.model(input_ids.to(self.dev), self.cache)
        self.token.copy_(logits[0, -1].argmax().to(torch.int32).reshape(1))
        self.pos.fill_(S)
        self._prompt_len = S
        self._steps = S
        self.first_token = int(self.token.item())
        return logits

    @torch.no_grad()
    def prime(self, input_ids: torch.Tensor) -> int:
        """The prompt into the KV cache through ``onebit_mixed_step`` (chunks of ``prefill_rows`` tokens, each ONE C call: GEMMs over
        all rows, ragged flash attention with past, lm_head + greedy token on the chunk's last row) and the engine armed with the
        first greedy token, which is returned.  Falls back to ``prefill`` when the native step does not take the model's shape
        (head_dim other than 64 / 128)."""
        if input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise ValueError("DecodeEngine is batch 1: input_ids must be [1, S]")
        S = input_ids.shape[1]
        if S < 1 or S + 1 > self.max_len:
            raise ValueError("prompt empty or longer than max_len")
        if self._native_prefill and self._mixed is None:
            try:
                self._mixed = MixedStep(self.model, self.cache.layers, 1, self.max_len, max_rows=min(S, self._prefill_rows))
            except ValueError:
                self._native_prefill = False
        if not self._native_prefill:
            self.prefill(input_ids)
            return self.first_token
        toks = input_ids[0].tolist()
        nxt = None
        for c0 in range(0, S, self._prefill_rows):
            nxt = self._mixed.launch([(0, c0, toks[c0:c0 + self._prefill_rows])])
        self.cache.length = S
        self.token.copy_(nxt[:1])
        self.pos.fill_(S)
        self._prompt_len = S
        self._steps = S
        self.first_token = int(self.token.item())
        return self.first_token

    def set_state(self, token: int, pos: int):
        """Arm the engine with ``token`` at cache position ``pos`` (the KV cache must hold ``pos`` tokens)."""
        if not 0 <= int(pos) < self.max_len:
            raise ValueError(f"set_state: position {pos} outside the KV cache [0, {self.max_len})")
        if not 0 <= int(token) < self.cfg.vocab_size:
            raise ValueError(f"set_state: token {token} outside the vocabulary")
        self.token.fill_(int(token))
        self.pos.fill_(int(pos))
        self._steps = int(pos)

    def step(self):
        """Decode one token (asynchronous): consumes the device-side token, appends to the KV cache,
        leaves the next greedy token on the device.  Raises when the KV cache is full: the kernels
        index the cache, the rope tables and their score buffers by the device-side position."""
        if self._steps >= self.max_len:
            raise RuntimeError(f"DecodeEngine.step: KV cache full ({self.max_len} positions); build the engine with a larger max_len")
        long = bool(self._long_from) and (self._steps >= self._long_from or not self._short_ok)
        b64 = not long and self._steps < 64
        if long and self._keyblock:
            ns = self._kb_splits(self._steps + 1)                  # this step attends to positions 0 .. _steps
            g = self._kb_graphs.get(ns)
            if g is not None:
                g.replay()
            else:
                self._launch(state=self._kb_states[ns][0])
            self._steps += 1
            return
        g = self.graph_long if long else (self.graph64 if b64 else self.graph)
        if g is not None:
            g.replay()
        else:
            self._launch(long=long, blind64=b64)
        self._steps += 1

    def logits(self) -> torch.Tensor:
        """fp32 logits of the last step (the reference returns logits.float())."""
        return self.buf["logits"].float()

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, max_new_tokens: int, eos_token_id=None) -> torch.Tensor:
        """Greedy decoding as the reference's ``greedy_search`` (generation/utils.py:2491-2540) for one
        sequence: the prompt followed by up to ``max_new_tokens`` tokens; with ``eos_token_id`` (int or
        list) the output ends with the first EOS token, as the reference's stopping criterion does
        (the steps are enqueued without a host round trip, the cut is made afterwards)."""
        S = input_ids.shape[1]
        if S + max_new_tokens > self.max_len:
            raise ValueError(f"generate: prompt {S} + max_new_tokens {max_new_tokens} exceeds the engine's max_len {self.max_len}")
        self.prime(input_ids)
        n = max_new_tokens - 1
        for _ in range(max(n, 0)):
            self.step()
        torch.cuda.synchronize(self.dev)
        new: List[int] = [self.first_token] if max_new_tokens > 0 else []
        if n > 0:
            new += self.out_tokens[self._prompt_len:self._prompt_len + n].tolist()
        if eos_token_id is not None:
            eos = set(eos_token_id) if isinstance(eos_token_id, (list, tuple, set)) else {int(eos_token_id)}
            for i, t in enumerate(new):
                if t in eos:
                    new = new[:i + 1]
                    break
        return torch.cat([input_ids.to(self.dev), torch.tensor([new], device=self.dev, dtype=input_ids.dtype)], dim=1)


class BatchedDecodeStep:
    """``onebit_decode_step_batched`` bound to a model and B KV-cache slots: one new token for every
    slot per call (skinny 1-bit GEMMs over the [B, hidden] rows, row-wise glue kernels, attention per
    (head, slot)); the caller owns scheduling, lm_head and sampling.  ``caches[l] = (k, v)`` with k, v
    ``[B, n_kv_heads, max_len, head_dim]`` fp16 on the model's device -- for an fp32 checkpoint build them with
    ``fp16_view(model).new_cache(B, max_len)`` (the model's own ``new_cache`` would be fp32 and is refused)."""

    def __init__(self, model: OneBitLlamaForCausalLM, caches, batch: int, max_len: int, sample: bool = True,
                 keep_logits: bool = False, producer_stats: bool = True, prescaled_rows: bool = True, chains: int = 0,
                 attn_splits: int = 0, attn_chunk: int = 256):
        """``attn_splits`` > 0: attention over key blocks (``onebit_attention_decode_rows``: ``attn_splits`` workgroups of
        ``attn_chunk`` positions per (head, slot)) -- any ``max_len``, and the form to use once contexts pass a few hundred
        positions; ``attn_splits * attn_chunk`` must cover the longest context + 1.  0: one workgroup per (head, slot)."""
        cfg = model.config
        if not model.lm_head.weight.is_cuda:
            raise RuntimeError("BatchedDecodeStep needs the model on a ROCm GPU (no CPU fallback)")
        model = fp16_view(model)        # an fp32 checkpoint: floating parameters cast once, packed weights shared
        p = model.lm_head.weight
        if not 2 <= batch <= 64:
            raise ValueError("batch must be in 2..64")
        if max_len > cfg.max_position_embeddings:
            raise ValueError("max_len exceeds max_position_embeddings (the rope tables have that many rows)")
        self.model, self.cfg, self.dev, self.batch = model, cfg, p.device, batch
        self.lib = _lib.load()
        dev, f16 = self.dev, torch.float16
        f16_t = torch.float16
        H, I, D = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
        shape = (batch, cfg.num_key_value_heads, max_len, D)
        for i, (kc, vc) in enumerate(caches):
            if tuple(kc.shape) != shape or tuple(vc.shape) != shape or not kc.is_contiguous() or not vc.is_contiguous():
                raise ValueError(f"cache {i} must be contiguous {shape}")
            # the attention kernels read and write the caches as fp16 on the model's device whatever the checkpoint's
            # dtype: an fp32 model's own new_cache() would be reinterpreted silently (results wrong, nothing out of bounds)
            if kc.dtype != f16_t or vc.dtype != f16_t:
                raise ValueError(f"cache {i} must be float16 (got {kc.dtype} / {vc.dtype}): build the caches with "
                                 "fp16_view(model).new_cache(batch, max_len)")
            if kc.device != p.device or vc.device != p.device:
                raise ValueError(f"cache {i} must be on the model's device {p.device}")
        self._model, self._keep = _model_struct(model, caches, max_len)
        z = lambda *n: torch.zeros(*n, dtype=f16, device=dev)
        Hq, Hkv = cfg.num_attention_heads * D, cfg.num_key_value_heads * D
        self.tokens = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.pos = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        self.buf = dict(hres0=z(batch, H), hres1=z(batch, H), x=z(batch, H), act=z(batch, I), u_q=z(batch, Hq),
                        u_k=z(batch, Hkv), u_v=z(batch, Hkv), attn_out=z(batch, Hq), u_o=z(batch, H),
                        u_gate=z(batch, I), u_up=z(batch, I), u_down=z(batch, H))
        b = self.buf
        # lm_head + greedy argmax inside the step (the fp16 lm_head streamed once for all rows); needs hidden % 64 == 0
        self.next_tokens = self.logits = None
        nt = lg = pv = pi = None
        if sample and H % 64 == 0:
            nparts = -(-cfg.vocab_size // 128) * 64
            self.next_tokens = torch.zeros(batch, dtype=torch.int32, device=dev)
            self._part_val = torch.zeros(nparts, dtype=torch.float32, device=dev)
            self._part_idx = torch.zeros(nparts, dtype=torch.int32, device=dev)
            nt, pv, pi = self.next_tokens.data_ptr(), self._part_val.data_ptr(), self._part_idx.data_ptr()
            if keep_logits:
                self.logits = torch.zeros(batch, cfg.vocab_size, dtype=f16, device=dev)
                lg = self.logits.data_ptr()
        self._state = _BatchState(ctypes.sizeof(_BatchState), batch, self.tokens.data_ptr(), self.pos.data_ptr(), b["hres0"].data_ptr(),
                                  b["hres1"].data_ptr(), b["x"].data_ptr(), b["act"].data_ptr(), b["u_q"].data_ptr(),
                                  b["u_k"].data_ptr(), b["u_v"].data_ptr(), b["attn_out"].data_ptr(), b["u_o"].data_ptr(),
                                  b["u_gate"].data_ptr(), b["u_up"].data_ptr(), b["u_down"].data_ptr(), nt, lg, pv, pi, None, None,
                                  int(chains), 0, 0, None, None)
        if attn_splits:
            if attn_chunk < 64 or attn_chunk % 64 or attn_splits < 1:
                raise ValueError("attn_chunk must be a positive multiple of 64, attn_splits >= 1")
            self._q_rows = z(batch, Hq)
            nb = int(self.lib.onebit_attention_decode_scratch_bytes(batch, cfg.num_attention_heads, int(attn_splits)))
            self._attn_scratch = torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)
            self._state.attn_splits, self._state.attn_chunk = int(attn_splits), int(attn_chunk)
            self._state.q_rows, self._state.attn_scratch = self._q_rows.data_ptr(), self._attn_scratch.data_ptr()
        # room for the consumers' pre-scaled rows fp16(x * input_factor): the projections then take the LDS-DMA skinny GEMM
        if prescaled_rows:
            self._x_scaled = torch.zeros(3, batch, H, dtype=f16, device=dev)
            self._state.x_scaled = self._x_scaled.data_ptr()
        # scratch for the q|k|v LayerNorm partials (published by the GEMM, combined by the attention workgroups)
        if producer_stats:
            nst = int(self.lib.onebit_batch_stats_floats(ctypes.byref(self._model), batch))
            self._qkv_stats = torch.zeros(max(nst, 1), dtype=torch.float32, device=dev)
            self._state.qkv_stats = self._qkv_stats.data_ptr()
        self.lib.onebit_decode_step_batched.restype = ctypes.c_int
        self.lib.onebit_decode_step_batched.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_BatchState), _vp]

    def launch(self) -> torch.Tensor:
        """Enqueue one step on the current stream; returns the final-norm output x [B, hidden]
        (a view of a persistent buffer).  With ``sample`` the greedy next token of every slot is in
        ``self.next_tokens`` (int32 [B]) afterwards, computed by the same call."""
        with torch.cuda.device(self.dev):
            rc = self.lib.onebit_decode_step_batched(ctypes.byref(self._model), ctypes.byref(self._state),
                                                     torch.cuda.current_stream(self.dev).cuda_stream)
        _lib.check(rc, "onebit_decode_step_batched")
        return self.buf["x"]


class MixedStep:
    """``onebit_mixed_step`` bound to a model and its KV-cache slots: ONE scheduler step of continuous batching (BASELINE config 5)
    on native kernels -- the token rows of all scheduled items concatenated, every 1-bit projection one GEMM over all rows, fused
    row glue, ragged attention (prompt chunks: causal flash attention with past; single-token rows: split-KV decode attention),
    lm_head + greedy token on the last row of every item.  ``caches[l] = (k, v)``, ``[n_slots, n_kv_heads, max_len, head_dim]``
    fp16.  The reference has no counterpart for the batching; per row the arithmetic is modeling_bitllama.py:869-918, 487-585."""

    def __init__(self, model: OneBitLlamaForCausalLM, caches, n_slots: int, max_len: int, max_rows: int = 4096,
                 attn_chunk: int = 256, keep_logits: bool = False):
        cfg = model.config
        if not model.lm_head.weight.is_cuda:
            raise RuntimeError("MixedStep needs the model on a ROCm GPU (no CPU fallback)")
        model = fp16_view(model)
        p = model.lm_head.weight
        H, D = cfg.hidden_size, cfg.head_dim
        if D not in (64, 128) or H % 64 != 0 or cfg.num_attention_heads * D != H:
            raise ValueError("MixedStep: head_dim must be 64 or 128 and hidden a multiple of 64")
        if max_len > cfg.max_position_embeddings:
            raise ValueError("max_len exceeds max_position_embeddings (the rope tables have that many rows)")
        self.model, self.cfg, self.dev = model, cfg, p.device
        self.n_slots, self.max_len, self.attn_chunk = int(n_slots), int(max_len), int(attn_chunk)
        if -(-self.max_len // self.attn_chunk) > 64:
            raise ValueError("MixedStep: attn_chunk too small for max_len (at most 64 splits)")
        shape = (n_slots, cfg.num_key_value_heads, max_len, D)
        for i, (kc, vc) in enumerate(caches):
            if tuple(kc.shape) != shape or tuple(vc.shape) != shape or not kc.is_contiguous() or not vc.is_contiguous() or \
                    kc.dtype != torch.float16 or vc.dtype != torch.float16 or kc.device != p.device or vc.device != p.device:
                raise ValueError(f"cache {i} must be contiguous float16 {shape} on {p.device}")
        self.lib = _lib.load()
        self._model, self._keep = _model_struct(model, caches, max_len)
        dev = self.dev
        nparts = -(-cfg.vocab_size // 128) * 64
        self._part_val = torch.zeros(nparts, dtype=torch.float32, device=dev)
        self._part_idx = torch.zeros(nparts, dtype=torch.int32, device=dev)
        self.keep_logits = keep_logits
        self.logits = None
        self.lib.onebit_mixed_workspace_bytes.argtypes = [ctypes.POINTER(_Model), _i64, _i32, _i32]
        self.lib.onebit_mixed_step.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_MixedState), _vp]
        self._rows = 0
        self._ensure(int(max_rows))
        self.launches = 0

    def _ensure(self, rows: int):
        """Room for a step of ``rows`` token rows (staging rows, workspace); grows geometrically, never shrinks."""
        if rows <= self._rows:
            return
        rows = max(rows, 2 * self._rows, 64)
        dev, ns = self.dev, self.n_slots
        nb = int(self.lib.onebit_mixed_workspace_bytes(ctypes.byref(self._model), rows, ns, self.attn_chunk))
        self._ws = None                                           # (free the old block first)
        self._ws = torch.zeros(nb + 256, dtype=torch.uint8, device=dev)
        self._ws_off = (-self._ws.data_ptr()) % 256
        # one pinned staging row [tokens | row_slot | row_pos | out_rows] and ONE asynchronous copy per step; two of them in turn,
        # each guarded by an event, so that a caller who enqueues steps without synchronising never rewrites a row in flight
        self._h_stage = [torch.zeros(4 * rows + ns, dtype=torch.int32).pin_memory() for _ in range(2)]
        self._h_np = [h.numpy() for h in self._h_stage]
        self._h_ev = [None, None]
        self._d_stage = [torch.zeros(4 * rows + ns, dtype=torch.int32, device=dev) for _ in range(2)]
        self.next_tokens = torch.zeros(rows + ns, dtype=torch.int32, device=dev)
        self._rows = rows
        self._ensure_logits(ns)

    def _ensure_logits(self, n: int):
        if self.keep_logits and (self.logits is None or self.logits.shape[0] < n):
            self.logits = None
            self.logits = torch.zeros(max(n, self.n_slots), self.cfg.vocab_size, dtype=torch.float16, device=self.dev)

    @torch.no_grad()
    def launch(self, items) -> torch.Tensor:
        """Enqueue one step on the current stream.  ``items``: ``(slot, start, tokens)`` per scheduled request (distinct slots):
        ``tokens`` enter the request's cache slot at positions ``start ...``.  Returns the device tensor of greedy next tokens,
        one per item in the order given (the token after the item's LAST row: for a prompt chunk that is not the prompt's last one
        the caller ignores it).  An item may carry a fourth element ``k`` (default 1): lm_head runs on its last ``k`` rows (the
        evaluation caller scores a request's continuation rows) and the item contributes ``k`` consecutive entries to the returned
        tokens and to ``logits``.  Asynchronous: synchronise before reading."""
        n_items = len(items)
        if n_items == 0 or n_items > self.n_slots:
            raise ValueError(f"MixedStep.launch: {n_items} items for {self.n_slots} slots")
        T = sum(len(it[2]) for it in items)
        self._ensure(T)
        R, ns, V = self._rows, self.n_slots, self.cfg.vocab_size
        sb = self.launches & 1
        if self._h_ev[sb] is not None:
            self._h_ev[sb].synchronize()
        st = self._h_np[sb]
        tok, rs, rp, outr = st[:R], st[R:2 * R], st[2 * R:3 * R], st[3 * R:]
        order = sorted(range(n_items), key=lambda i: len(items[i][2]) != 1)          # single-token rows first (stable)
        outs = [int(it[3]) if len(it) > 3 else 1 for it in items]
        out0 = [0] * n_items
        for i in range(1, n_items):
            out0[i] = out0[i - 1] + outs[i - 1]
        n_out = out0[-1] + outs[-1]
        self._ensure_logits(n_out)
        segs, row, n_dec, dec_ctx, seen = [], 0, 0, 0, set()
        for i in order:
            slot, start, toks = items[i][:3]
            n = len(toks)
            if not 1 <= outs[i] <= n:
                raise ValueError(f"MixedStep.launch: item {i} asks for {outs[i]} output rows of {n}")
            if n < 1 or not 0 <= slot < ns or start < 0 or start + n > self.max_len or slot in seen:
                raise ValueError(f"MixedStep.launch: item {i} (slot {slot}, start {start}, {n} tokens) outside the cache or a repeated slot")
            if min(toks) < 0 or max(toks) >= V:
                raise ValueError(f"MixedStep.launch: item {i} has a token outside the vocabulary")
            seen.add(slot)
            tok[row:row + n] = toks
            rs[row:row + n] = slot
            rp[row:row + n] = range(start, start + n)
            if n == 1:
                n_dec += 1
                dec_ctx = max(dec_ctx, start + 1)
            else:
                segs.append((row, n, slot, start))
            outr[out0[i]:out0[i] + outs[i]] = range(row + n - outs[i], row + n)
            row += n
        d = self._d_stage[sb]
        d.copy_(self._h_stage[sb], non_blocking=True)                                # stream-ordered before the kernels
        self._h_ev[sb] = torch.cuda.Event()
        self._h_ev[sb].record(torch.cuda.current_stream(self.dev))
        seg_arr = (_Seg * max(len(segs), 1))(*[_Seg(*g) for g in segs])
        base = d.data_ptr()
        state = _MixedState(ctypes.sizeof(_MixedState), T, n_dec, len(segs), n_out, ns, self.attn_chunk, dec_ctx,
                            base, base + 4 * R, base + 8 * R, seg_arr, base + 12 * R, self.next_tokens.data_ptr(),
                            None if self.logits is None else self.logits.data_ptr(), self._part_val.data_ptr(), self._part_idx.data_ptr(),
                            self._ws.data_ptr() + self._ws_off, self._ws.numel() - self._ws_off)
        with torch.cuda.device(self.dev):
            rc = self.lib.onebit_mixed_step(ctypes.byref(self._model), ctypes.byref(state), torch.cuda.current_stream(self.dev).cuda_stream)
        _lib.check(rc, "onebit_mixed_step")
        self.launches += 1
        return self.next_tokens[:n_out]
