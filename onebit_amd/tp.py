"""Model-level tensor-parallel prefill over RCCL -- SURVEY.md section 8(e), BASELINE configs 3/4
("prefill shards the hidden dim across up to 8 GPUs").

The seven 1-bit call sites of a decoder layer (modeling_bitllama.py:229-231, 257, 451-454, 522-524,
580) are paired Megatron-style on the sharded layer forms of ``onebit_amd.sharded``:

    x  = RMSNorm(h)                               h, x: all T tokens on every rank
    q, k, v   N-sharded by head                   rank p owns heads [p H/n, (p+1) H/n): its rows of the packed
                                                  matrices, g; the LayerNorm over the complete row needs
                                                  2 floats per token and projection -> one all_gather of
                                                  [T, 2] fp32 per projection (NOT the activations)
    attention on the local heads                  no communication (RoPE, causal softmax, P.V)
    o         K-sharded on the local heads' columns   fp32 partial sums [T, hidden] -> reduce_scatter over tokens
                                                  -> g, LayerNorm on the rank's own T/n COMPLETE rows
    h_own += o_own;  x2_own = RMSNorm(h_own)      the residual stream lives token-sharded from here
    all_gather(x2_own) -> x2 [T, hidden] fp16     (with the reduce_scatter: exchange 1)
    gate, up  N-sharded                           two [T, 2] statistics all_gathers
    act = silu(gate) * up on the local columns
    down      K-sharded on the local columns      partial sums -> reduce_scatter -> g, LayerNorm on own rows
    h_own += down_own;  all_gather(RMSNorm_next(h_own))   (exchange 2)

Two activation exchanges per layer instead of seven (one per BitLinearInf under plain K-sharding), each a
reduce_scatter of fp32 [T, hidden] plus an all_gather of fp16 [T, hidden]; the packed weights and the
KV cache are split n ways.  xGMI is point-to-point: at T = 16384, hidden 4096 an exchange moves
268 MB (fp32) + 134 MB (fp16) per rank pair-wise, against ~25 ms of per-rank MFMA time per layer at
n = 8 -- the design point where prefill starts to scale.

Compute goes through callbacks with the contracts of ``onebit_amd.sharded`` (HIP kernels through the C
ABI by default; the gloo CPU tests pass oracle-backed ones), so the control flow tested on CPU is the
one that runs over RCCL.

``fused=True`` (the default with the HIP callbacks) runs the row-wise glue between the sharded GEMMs through the
same fused kernels as the single-GPU prefill route (``OneBitLlamaForCausalLM.set_fused_glue``) instead of torch
elementwise ops and separate statistics / normalise passes:

    q|k|v   pre-LayerNorm rows of the local heads -> local row statistics (one pass each) -> all_gather [T, 2] x 3 ->
            ``onebit_rows_qkv_rope_stats``: LayerNorm with the COMBINED statistics + RoPE + head transpose, k / v
            straight into the rank's KV rows, q token-major
    o, down fp32 partials -> reduce_scatter -> ``onebit_scale_layernorm(SKIP_LN)`` (u = fp16(fp16(z) * g) on the own
            rows) -> ``onebit_rows_res_ln_rms``: LayerNorm + residual + RMSNorm of the NEXT block in one pass
    gate|up local row statistics -> all_gather -> ``onebit_rows_swiglu_stats``: LayerNorm x 2 + SiLU * up in one pass

(degree 1 on one MI355X, 7B, 8 x 2048 tokens: 308 ms with the torch glue).  The fused steps are callbacks too
(``glue=``), so the gloo tests run this exact control flow with torch stand-ins.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import sharded
from .llama import OneBitLlamaForCausalLM, _rotate_half
from .sharded import KShard, NShard

__all__ = ["TPPlan", "TensorParallelPrefill", "HipGlue"]


class HipGlue:
    """The fused row kernels of the tensor-parallel route through the C ABI (fp16, no projection bias)."""

    @staticmethod
    def qkv_rope(u_q, u_k, u_v, stats6, cos, sin, B, S, Hl, Hkvl, D, eps):
        """LayerNorm (given statistics) + RoPE + head transpose: q [B, S, Hl, D] (token-major), k, v [B, Hkvl, S, D]."""
        from . import _lib
        from .bitnet import _stream_ptr
        lib = _lib.load()
        dev, dt = u_q.device, u_q.dtype
        q = torch.empty((B, S, Hl, D), dtype=dt, device=dev)
        k = torch.empty((B, Hkvl, S, D), dtype=dt, device=dev)
        v = torch.empty((B, Hkvl, S, D), dtype=dt, device=dev)
        st = stats6.contiguous()
        with torch.cuda.device(dev):
            rc = lib.onebit_rows_qkv_rope_stats(u_q.data_ptr(), u_k.data_ptr(), u_v.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                st.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), B, S, Hl, Hkvl, D, 0, S,
                                                cos.shape[0], eps, _lib.FLAG_Q_TOKEN_MAJOR, _stream_ptr(dev))
        _lib.check(rc, "onebit_rows_qkv_rope_stats")
        return q, k, v

    @staticmethod
    def swiglu(u_g, u_u, stats4, eps, h_next=None):
        from . import _lib
        from .bitnet import _stream_ptr
        lib = _lib.load()
        act = torch.empty_like(u_g)
        st = stats4.contiguous()
        with torch.cuda.device(u_g.device):
            rc = lib.onebit_rows_swiglu_stats(u_g.data_ptr(), u_u.data_ptr(), None if h_next is None else h_next.data_ptr(),
                                              st.data_ptr(), act.data_ptr(), u_g.shape[0],
                                              u_g.shape[1], eps, _stream_ptr(u_g.device))
        _lib.check(rc, "onebit_rows_swiglu_stats")
        return act

    @staticmethod
    def u_rows(shard: KShard, z, dtype):
        """u = fp16(fp16(z) * g) of complete rows (bitnet.py:115-116), LayerNorm left to the consumer."""
        from . import _lib
        from .bitnet import _dtype_code, _stream_ptr
        lib = _lib.load()
        u = torch.empty(z.shape, dtype=dtype, device=z.device)
        g = shard.weight_scale.to(dtype)
        with torch.cuda.device(z.device):
            rc = lib.onebit_scale_layernorm(z.data_ptr(), g.data_ptr(), None, u.data_ptr(), None, z.shape[0], z.shape[1],
                                            _dtype_code(dtype), 0.0, _lib.FLAG_SKIP_LN, _stream_ptr(z.device))
        _lib.check(rc, "onebit_scale_layernorm")
        return u

    @staticmethod
    def res_ln_rms(h, u, w, rms_eps, ln_eps, h_next=()):
        """h + LayerNorm(u) -> new residual rows; RMSNorm(that) * w -> x.  ``h_next`` (<= 3 input_factor vectors): also
        the consumers' pre-scaled rows fp16(x * h_i) (returned as a list; x itself is then not written)."""
        import ctypes
        from . import _lib
        from .bitnet import _stream_ptr
        lib = _lib.load()
        hout = torch.empty_like(h)
        x = None if h_next else torch.empty_like(h)
        xs = [torch.empty_like(h) for _ in h_next]
        hp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in h_next])
        xp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in xs])
        with torch.cuda.device(h.device):
            rc = lib.onebit_rows_res_ln_rms(h.data_ptr(), u.data_ptr(), w.data_ptr(), hout.data_ptr(), None if x is None else x.data_ptr(),
                                            hp, xp, len(xs), h.shape[0], h.shape[1], rms_eps, ln_eps, _stream_ptr(h.device))
        _lib.check(rc, "onebit_rows_res_ln_rms")
        return (hout, x) if not h_next else (hout, xs)

    @staticmethod
    def attention(q, k, v, impl="hip", h_next=None):
        """Causal attention on the local heads: q [B, S, Hl, D] token-major, k / v [B, Hkvl, S, D] -> [B * S, Hl * D].
        ``impl="hip"``: the build's own flash kernel (onebit_attention_prefill; with ``h_next`` the rows come back already
        multiplied by o_proj's input_factor slice); "sdpa": torch's fused attention."""
        B, S, Hl, D = q.shape
        if impl == "hip" and D in (64, 128) and q.dtype == torch.float16:
            from .llama import hip_attention_prefill
            return hip_attention_prefill(q, k, v, 0, h_next).view(B * S, Hl * D)
        if h_next is not None:
            raise ValueError("pre-scaled attention rows need the hip attention kernel")
        if k.shape[1] != Hl:
            k, v = k.repeat_interleave(Hl // k.shape[1], dim=1), v.repeat_interleave(Hl // v.shape[1], dim=1)
        o = nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k, v, is_causal=True)
        return o.transpose(1, 2).contiguous().reshape(B * S, Hl * D)        # (no copy: the output is token-major)


@dataclass
class _LayerShards:
    q: NShard
    k: NShard
    v: NShard
    o: KShard
    gate: NShard
    up: NShard
    down: KShard


def _rows(mod, n0: int, n1: int) -> NShard:
    b = None if mod.bias is None else mod.bias.data[n0:n1].contiguous()
    return NShard(mod.weight.data[n0:n1], mod.input_factor.data, mod.weight_scale.data[n0:n1].contiguous(), b,
                  n0, n1, mod.in_features, mod.out_features)


def _cols(mod, k0: int, k1: int) -> KShard:
    if k0 % 32 or k1 % 32:
        raise ValueError("K slices must fall on packed-dword boundaries (multiples of 32)")
    b = None if mod.bias is None else mod.bias.data
    return KShard(mod.weight.data[:, k0 // 8:k1 // 8].contiguous(), mod.input_factor.data[k0:k1].contiguous(),
                  mod.weight_scale.data, b, k0, k1, mod.in_features, mod.out_features)


@dataclass
class TPPlan:
    """Which heads / intermediate columns a rank owns."""
    rank: int
    world: int
    heads: range          # query heads
    kv_heads: range
    inter: range          # intermediate columns (multiple-of-32 boundaries)

    @staticmethod
    def make(cfg, rank: int, world: int) -> "TPPlan":
        H, Hkv, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
        if H % world or Hkv % world:
            raise ValueError(f"heads ({H} query, {Hkv} kv) must divide by the tensor-parallel degree {world}")
        if (cfg.head_dim * (H // world)) % 32:
            raise ValueError("a rank's head columns must be a multiple of 32 (o_proj K slice)")
        i0, i1 = sharded.n_range(I, rank, world, granule=32)
        return TPPlan(rank, world, range(rank * H // world, (rank + 1) * H // world),
                      range(rank * Hkv // world, (rank + 1) * Hkv // world), range(i0, i1))


class TensorParallelPrefill:
    """Prefill of ``model`` with its 1-bit layers split over the ranks of ``group``.

    ``forward(input_ids [B, S])`` returns the fp32 logits of this rank's token rows (``own_logits``)
    and, with ``gather_logits=True``, the complete ``[B, S, vocab]`` tensor on every rank.  KV of the
    local heads is kept per layer in ``self.kv`` (``[B, local_kv_heads, S, D]``) for a following
    tensor-parallel decode."""

    def __init__(self, model: OneBitLlamaForCausalLM, rank: int, world: int, group=None,
                 rows_fn: Callable = sharded.hip_rows_u, stats_fn: Callable = sharded.hip_row_stats,
                 normalize_fn: Callable = sharded.hip_normalize, partial_fn: Callable = sharded.hip_partial,
                 epilogue_fn: Callable = sharded.hip_epilogue, attention: str = "eager", fused: Optional[bool] = None,
                 glue=None):
        self.model, self.cfg, self.group = model, model.config, group
        hip_default = (rows_fn is sharded.hip_rows_u and stats_fn is sharded.hip_row_stats and partial_fn is sharded.hip_partial)
        D_ = model.config.head_dim
        can_fuse = (glue is not None) or (hip_default and model.lm_head.weight.dtype == torch.float16 and D_ >= 16 and D_ & (D_ - 1) == 0
                                         and all(p_.bias is None for l_ in model.model.layers
                                                 for p_ in (l_.self_attn.q_proj, l_.self_attn.k_proj, l_.self_attn.v_proj,
                                                            l_.self_attn.o_proj, l_.mlp.gate_proj, l_.mlp.up_proj, l_.mlp.down_proj)))
        if fused and not can_fuse:
            raise ValueError("fused tensor-parallel glue needs an fp16 model without projection biases and a power-of-two head_dim")
        # default: fused whenever it is possible AND the caller asked for the fused attention (attention="eager" keeps the
        # reference's op order end to end, torch glue included)
        self.fused = (can_fuse and (attention in ("sdpa", "hip") or glue is not None)) if fused is None else bool(fused)
        self.glue = glue if glue is not None else HipGlue
        self.plan = TPPlan.make(self.cfg, rank, world)
        self.fns = dict(rows_fn=rows_fn, stats_fn=stats_fn, normalize_fn=normalize_fn)
        self.partial_fn, self.epilogue_fn = partial_fn, epilogue_fn
        self.attention = attention
        D = self.cfg.head_dim
        p = self.plan
        self.layers: List[_LayerShards] = []
        for layer in model.model.layers:
            a, m = layer.self_attn, layer.mlp
            self.layers.append(_LayerShards(
                q=_rows(a.q_proj, p.heads.start * D, p.heads.stop * D),
                k=_rows(a.k_proj, p.kv_heads.start * D, p.kv_heads.stop * D),
                v=_rows(a.v_proj, p.kv_heads.start * D, p.kv_heads.stop * D),
                o=_cols(a.o_proj, p.heads.start * D, p.heads.stop * D),
                gate=_rows(m.gate_proj, p.inter.start, p.inter.stop),
                up=_rows(m.up_proj, p.inter.start, p.inter.stop),
                down=_cols(m.down_proj, p.inter.start, p.inter.stop)))
        self.kv: List = []
        self.exchanges = 0          # reduce_scatter + all_gather pairs issued by the last forward

    # ---- collectives -------------------------------------------------------------------------------
    def _world(self) -> int:
        return self.plan.world

    def _reduce_scatter_rows(self, zp: torch.Tensor, rows: int) -> torch.Tensor:
        """fp32 partial sums [Tp, N] (Tp = rows * world) -> this rank's `rows` complete rows."""
        if self._world() == 1:
            return zp
        mine = torch.empty((rows, zp.shape[1]), dtype=zp.dtype, device=zp.device)
        dist.reduce_scatter_tensor(mine, zp, op=dist.ReduceOp.SUM, group=self.group)
        return mine

    def _all_gather_rows(self, own: torch.Tensor) -> torch.Tensor:
        if self._world() == 1:
            return own
        full = torch.empty((own.shape[0] * self._world(), own.shape[1]), dtype=own.dtype, device=own.device)
        dist.all_gather_into_tensor(full, own.contiguous(), group=self.group)
        return full

    def _n_sharded(self, shard: NShard, x: torch.Tensor) -> torch.Tensor:
        # the statistics exchange needs the per-rank row counts: equal slices by construction here
        return _n_sharded_equal(shard, x, self.group, self._world(), **self.fns)

    # ---- forward -----------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, gather_logits: bool = True):
        if self.fused:
            return self._forward_fused(input_ids, gather_logits)
        cfg, model, p = self.cfg, self.model, self.plan
        n = self._world()
        B, S = input_ids.shape
        T = B * S
        rows = -(-T // n)                       # token rows per rank (padded)
        Tp = rows * n
        D, Hl, Hkvl = cfg.head_dim, len(p.heads), len(p.kv_heads)
        emb = model.model.embed_tokens(input_ids).reshape(T, -1)
        dev, dt = emb.device, emb.dtype
        cos, sin = model._rope_tables(dev, dt)
        cos, sin = cos[:S][None, None], sin[:S][None, None]
        pad = lambda t: t if t.shape[0] == Tp else torch.cat([t, t.new_zeros(Tp - t.shape[0], t.shape[1])], dim=0)
        h_own = pad(emb)[p.rank * rows:(p.rank + 1) * rows].clone()        # residual stream: own token rows
        x = pad(model.model.layers[0].input_layernorm(emb)) if len(model.model.layers) else None
        self.kv, self.exchanges = [], 0
        for li, (layer, sh) in enumerate(zip(model.model.layers, self.layers)):
            # --- attention block: q, k, v of the local heads for ALL tokens
            q = self._n_sharded(sh.q, x)[:T].view(B, S, Hl, D).transpose(1, 2)
            k = self._n_sharded(sh.k, x)[:T].view(B, S, Hkvl, D).transpose(1, 2)
            v = self._n_sharded(sh.v, x)[:T].view(B, S, Hkvl, D).transpose(1, 2)
            q = (q * cos) + (_rotate_half(q) * sin)                      # apply_rotary_pos_emb, modeling_bitllama.py:175-181
            k = (k * cos) + (_rotate_half(k) * sin)
            self.kv.append((k, v))
            keys, vals = k, v
            if Hkvl != Hl:
                keys = keys.repeat_interleave(Hl // Hkvl, dim=1)
                vals = vals.repeat_interleave(Hl // Hkvl, dim=1)
            if self.attention == "sdpa":
                o = nn.functional.scaled_dot_product_attention(q, keys, vals, is_causal=True)
            else:
                w = torch.matmul(q, keys.transpose(2, 3)) / math.sqrt(D)                 # :546
                if S > 1:
                    w = w + torch.triu(torch.full((S, S), torch.finfo(w.dtype).min, device=dev, dtype=w.dtype), diagonal=1)[None, None]
                w = nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)    # :562
                o = torch.matmul(w, vals)
            o = pad(o.transpose(1, 2).reshape(T, Hl * D))
            # --- o_proj on the local heads' columns: partial sums, ONE reduction, epilogue on own rows
            zo = self._reduce_scatter_rows(self.partial_fn(sh.o, o), rows)
            h_own = h_own + self.epilogue_fn(sh.o, zo, dt, 1e-5)
            x2 = self._all_gather_rows(layer.post_attention_layernorm(h_own))
            self.exchanges += 1
            # --- MLP: gate, up on the local columns, down K-sharded on them
            g = self._n_sharded(sh.gate, x2)
            u = self._n_sharded(sh.up, x2)
            act = nn.functional.silu(g) * u                                               # :257
            zd = self._reduce_scatter_rows(self.partial_fn(sh.down, act), rows)
            h_own = h_own + self.epilogue_fn(sh.down, zd, dt, 1e-5)
            self.exchanges += 1
            if li + 1 < len(self.layers):
                x = self._all_gather_rows(model.model.layers[li + 1].input_layernorm(h_own))
        own_logits = model.lm_head(model.model.norm(h_own)).float()
        if not gather_logits:
            return own_logits
        logits = self._all_gather_rows(own_logits)[:T]
        return logits.view(B, S, -1)

    def _complete_stats(self, parts: List[torch.Tensor], shards: List[NShard], eps: float) -> torch.Tensor:
        """Local {mean, M2} of several N-sharded row blocks -> {mean, rstd} of the COMPLETE rows, [T, 2 * len(parts)]
        (one all_gather for all of them; parallel-variance combine, Chan et al.)."""
        world = self._world()
        st = torch.stack(parts, dim=0)                                           # [P, T, 2]
        P, T = st.shape[0], st.shape[1]
        # (the row counts are constants of the plan: built once per (shapes, device) -- a torch.tensor(..., device=) here
        #  was a blocking host -> device copy twice per layer)
        key = (tuple((sh.out_features, sh.n1 - sh.n0) for sh in shards), st.device)
        cache = self.__dict__.setdefault("_stat_consts", {})
        if key not in cache:
            cache[key] = (torch.tensor([float(sh.out_features) for sh in shards], device=st.device).view(P, 1),
                          torch.tensor([float(sh.n1 - sh.n0) for sh in shards], device=st.device))
        Ns, cnt_c = cache[key]
        if world > 1:
            cnt = cnt_c
            all_st = torch.empty((world,) + tuple(st.shape), dtype=st.dtype, device=st.device)
            all_cnt = torch.empty((world, P), dtype=torch.float32, device=st.device)
            dist.all_gather_into_tensor(all_st.view(world * P * T, 2), st.contiguous().view(P * T, 2), group=self.group)
            dist.all_gather_into_tensor(all_cnt.view(world * P), cnt, group=self.group)
            c = all_cnt.view(world, P, 1)
            mean = (all_st[..., 0] * c).sum(dim=0) / Ns                          # [P, T]
            m2 = all_st[..., 1].sum(dim=0) + (c * (all_st[..., 0] - mean[None]) ** 2).sum(dim=0)
        else:
            mean, m2 = st[..., 0], st[..., 1]
        rstd = torch.rsqrt(m2 / Ns + eps)
        return torch.stack([mean, rstd], dim=-1).permute(1, 0, 2).reshape(T, 2 * P).contiguous()

    @torch.no_grad()
    def _forward_fused(self, input_ids: torch.Tensor, gather_logits: bool = True):
        cfg, model, p, G = self.cfg, self.model, self.plan, self.glue
        n = self._world()
        B, S = input_ids.shape
        T = B * S
        rows = -(-T // n)
        Tp = rows * n
        D, Hl, Hkvl = cfg.head_dim, len(p.heads), len(p.kv_heads)
        rows_fn, stats_fn = self.fns["rows_fn"], self.fns["stats_fn"]
        emb = model.model.embed_tokens(input_ids).reshape(T, -1)
        dev, dt = emb.device, emb.dtype
        cos, sin = model._rope_tables(dev, dt)
        pad = lambda t: t if t.shape[0] == Tp else torch.cat([t, t.new_zeros(Tp - t.shape[0], t.shape[1])], dim=0)
        h_own = pad(emb)[p.rank * rows:(p.rank + 1) * rows].contiguous()
        layers = model.model.layers
        x = pad(layers[0].input_layernorm(emb)) if len(layers) else None
        x_own = None
        self.kv, self.exchanges = [], 0
        # Degree 1 with the HIP kernels: nothing is exchanged, so the producers can hand the consumers their pre-scaled rows
        # fp16(x * input_factor) directly (ONEBIT_FLAG_PRESCALED, as the single-GPU fused route does) and a K-"shard" is the
        # whole layer: its fp16 pre-LayerNorm rows come straight from the GEMM instead of fp32 partials + a rounding pass.
        # With n > 1 the gathered x is shared by consumers with different input_factors and the partial sums cross ranks.
        direct = n == 1 and G is HipGlue and rows_fn is sharded.hip_rows_u and self.partial_fn is sharded.hip_partial
        pre = lambda shards: direct and all(sharded.hip_prescaled_ok(s_, Tp, dev) for s_ in shards)
        xs = None                                   # pre-scaled copies of x for the next consumers, when available

        def k_rows(shard, a, a_scaled):
            """u (fp16, own rows) of a K-sharded layer: partial sums -> reduce_scatter -> fp16(fp16(z) * g); degree 1: direct."""
            if direct:
                full = NShard(shard.weight, shard.input_factor, shard.weight_scale, None, 0, shard.out_features,
                              shard.k1 - shard.k0, shard.out_features)
                return rows_fn(full, a, prescaled=True) if a_scaled else rows_fn(full, a)
            return G.u_rows(shard, self._reduce_scatter_rows(self.partial_fn(shard, a), rows), dt)

        for li, (layer, sh) in enumerate(zip(layers, self.layers)):
            # --- q | k | v of the local heads for all tokens; LayerNorm over the COMPLETE rows via combined statistics
            # (u and its local row statistics in one call: out of the GEMM epilogue where the shape allows, sharded.rows_and_stats)
            if xs is not None:
                (u_q, s_q), (u_k, s_k), (u_v, s_v) = (sharded.rows_and_stats(s_, a_, rows_fn, stats_fn, prescaled=True)
                                                      for s_, a_ in zip((sh.q, sh.k, sh.v), xs))
            else:
                (u_q, s_q), (u_k, s_k), (u_v, s_v) = (sharded.rows_and_stats(s_, x, rows_fn, stats_fn) for s_ in (sh.q, sh.k, sh.v))
            st6 = self._complete_stats([s_q, s_k, s_v], [sh.q, sh.k, sh.v], 1e-5)
            q, k, v = G.qkv_rope(u_q[:T], u_k[:T], u_v[:T], st6[:T], cos, sin, B, S, Hl, Hkvl, D, 1e-5)
            self.kv.append((k, v))
            # --- o_proj on the local heads' columns -> ONE reduction -> u on own rows -> LayerNorm + residual + RMSNorm fused
            if G is HipGlue:
                hip_attn = self.attention == "hip" and D in (64, 128) and dt == torch.float16
                o_pre = hip_attn and direct and T == Tp and sharded.hip_prescaled_ok(
                    NShard(sh.o.weight, sh.o.input_factor, sh.o.weight_scale, None, 0, sh.o.out_features, sh.o.k1 - sh.o.k0, sh.o.out_features), Tp, dev)
                o = pad(G.attention(q, k, v, "hip" if hip_attn else "sdpa", sh.o.input_factor if o_pre else None))
            else:
                o_pre = False
                o = pad(G.attention(q, k, v))
            u_o = k_rows(sh.o, o, o_pre)
            if pre((sh.gate, sh.up)):
                h_own, (ag, au) = G.res_ln_rms(h_own, u_o, layer.post_attention_layernorm.weight, cfg.rms_norm_eps, 1e-5,
                                               h_next=(sh.gate.input_factor, sh.up.input_factor))
                (u_g, s_g), (u_u, s_u) = (sharded.rows_and_stats(s_, a_, rows_fn, stats_fn, prescaled=True) for s_, a_ in ((sh.gate, ag), (sh.up, au)))
            else:
                h_own, x2_own = G.res_ln_rms(h_own, u_o, layer.post_attention_layernorm.weight, cfg.rms_norm_eps, 1e-5)
                x2 = self._all_gather_rows(x2_own)
                (u_g, s_g), (u_u, s_u) = (sharded.rows_and_stats(s_, x2, rows_fn, stats_fn) for s_ in (sh.gate, sh.up))
            self.exchanges += 1
            # --- MLP
            st4 = self._complete_stats([s_g, s_u], [sh.gate, sh.up], 1e-5)
            down_pre = direct and sharded.hip_prescaled_ok(NShard(sh.down.weight, sh.down.input_factor, sh.down.weight_scale, None, 0,
                                                                  sh.down.out_features, sh.down.k1 - sh.down.k0, sh.down.out_features), Tp, dev)
            act = G.swiglu(u_g, u_u, st4, 1e-5, sh.down.input_factor) if down_pre else G.swiglu(u_g, u_u, st4, 1e-5)
            u_d = k_rows(sh.down, act, down_pre)
            last = li + 1 >= len(layers)
            nxt = model.model.norm.weight if last else layers[li + 1].input_layernorm.weight
            nsh = None if last else self.layers[li + 1]
            if nsh is not None and pre((nsh.q, nsh.k, nsh.v)):
                h_own, xs = G.res_ln_rms(h_own, u_d, nxt, cfg.rms_norm_eps, 1e-5,
                                         h_next=(nsh.q.input_factor, nsh.k.input_factor, nsh.v.input_factor))
            else:
                xs = None
                h_own, x_own = G.res_ln_rms(h_own, u_d, nxt, cfg.rms_norm_eps, 1e-5)
                if not last:
                    x = self._all_gather_rows(x_own)
            self.exchanges += 1
        if x_own is None:                                                       # a model without layers
            x_own = model.model.norm(h_own)
        own_logits = model.lm_head(x_own).float()
        if not gather_logits:
            return own_logits
        return self._all_gather_rows(own_logits)[:T].view(B, S, -1)

    __call__ = forward


def _n_sharded_equal(shard: NShard, x: torch.Tensor, group, world: int, rows_fn, stats_fn, normalize_fn, eps: float = 1e-5):
    """``sharded.n_sharded_forward`` for an arbitrary (not n_range-derived) equal row split: every rank
    holds ``n1 - n0`` rows, the LayerNorm runs over all ``out_features`` of them."""
    u = rows_fn(shard, x)
    st = stats_fn(u)
    N, n_p = shard.out_features, shard.n1 - shard.n0
    if world > 1:
        all_st = torch.empty((world,) + tuple(st.shape), dtype=st.dtype, device=st.device)
        dist.all_gather_into_tensor(all_st.view(world * st.shape[0], 2), st.contiguous(), group=group)
        counts = torch.empty((world, 1), dtype=torch.float32, device=st.device)
        cnt = torch.tensor([float(n_p)], device=st.device)
        dist.all_gather_into_tensor(counts.view(world), cnt, group=group)            # slices may differ (intermediate / 32 units)
        mean = (all_st[:, :, 0] * counts).sum(dim=0) / N
        m2 = all_st[:, :, 1].sum(dim=0) + (counts * (all_st[:, :, 0] - mean[None]) ** 2).sum(dim=0)
    else:
        mean, m2 = st[:, 0].contiguous(), st[:, 1].contiguous()
    rstd = torch.rsqrt(m2 / N + eps)
    return normalize_fn(u, mean.contiguous(), rstd.contiguous(), shard.bias)
