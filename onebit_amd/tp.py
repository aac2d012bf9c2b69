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

__all__ = ["TPPlan", "TensorParallelPrefill"]


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
                 epilogue_fn: Callable = sharded.hip_epilogue, attention: str = "eager"):
        self.model, self.cfg, self.group = model, model.config, group
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
