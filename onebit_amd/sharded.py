"""K-sharded (hidden-dim sharded) 1-bit linear layer over RCCL -- SURVEY.md section 8(e).

``z = W+- . (h * x)`` is linear in K, so rank p keeps the column slice ``W[:, K_p]`` of the packed
matrix (a byte-column slice of the reference's int8 ``[N, K/8]`` tensor: the C ABI takes a row
pitch, nothing is repacked), ``h[K_p]`` and its slice of the activations, and produces fp32
partial sums ``z_p [T, N]`` (``onebit_matmul_partial``).  One exchange step follows -- and it has
to sit BEFORE the LayerNorm, whose statistics need complete rows:

  mode "allreduce":  all_reduce(sum) of z_p, then every rank applies g + LayerNorm to all T rows
  mode "rs_ag":      reduce_scatter over tokens -> each rank applies g + LayerNorm
                     (``onebit_scale_layernorm``) to its T/n complete rows -> all_gather of the
                     fp16 result.  Same bytes on the wire for the reduction, half the bytes for
                     the gather (fp16 instead of fp32), and the epilogue is done once, not n times.

On MI355X the backend "nccl" is RCCL over xGMI; tests run the identical control flow on CPU with
"gloo" and oracle-provided compute callbacks.  Decode (T = 1) stays single-GPU (BASELINE.json):
280 latency-bound 20-55 KB collectives per token cost more than they save.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["KShard", "shard_k", "k_sharded_forward", "hip_partial", "hip_epilogue", "KShardedBitLinear", "FusedKShardedDecoder", "lockstep_step",
           "shard_model_k", "NShard", "n_range", "shard_n", "n_sharded_forward", "hip_rows_u", "hip_row_stats", "hip_normalize"]


@dataclass
class KShard:
    """Rank-local slice of one BitLinearInf: packed[:, k0/8:k1/8] (a view), h[k0:k1], full g / bias."""
    weight: torch.Tensor          # int8 [N, (k1-k0)/8], a strided VIEW of the full packed matrix or a copy
    input_factor: torch.Tensor    # [k1-k0]
    weight_scale: torch.Tensor    # [N]
    bias: Optional[torch.Tensor]
    k0: int
    k1: int
    in_features: int
    out_features: int


def k_range(K: int, rank: int, world: int, granule: int = 32) -> Tuple[int, int]:
    """Contiguous K slice of `rank`, boundaries on multiples of `granule` (32 = one packed dword;
    4096, 11008, 5120, 13824 all split evenly into 2/4/8 such slices)."""
    units = K // granule
    if K % granule:
        raise ValueError(f"in_features={K} is not a multiple of {granule}")
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return u0 * granule, u1 * granule


def shard_k(weight: torch.Tensor, input_factor: torch.Tensor, weight_scale: torch.Tensor,
            bias: Optional[torch.Tensor], rank: int, world: int, copy: bool = True) -> KShard:
    N, KB = weight.shape
    K = KB * 8
    k0, k1 = k_range(K, rank, world)
    w = weight[:, k0 // 8:k1 // 8]
    if copy:
        w = w.contiguous()
    return KShard(w, input_factor[k0:k1].contiguous(), weight_scale, bias, k0, k1, K, N)


def hip_partial(shard: KShard, x_slice: torch.Tensor) -> torch.Tensor:
    """fp32 partial sums [T, N] of this rank's K slice through the C ABI (MFMA kernels)."""
    from . import _lib
    from .bitnet import _dtype_code, _stream_ptr
    lib = _lib.load()
    T, Ks = x_slice.shape
    zp = torch.empty((T, shard.out_features), dtype=torch.float32, device=x_slice.device)
    w = shard.weight
    code = _dtype_code(x_slice.dtype)
    # room for the pre-scaled slice: large calls then take the LDS-DMA GEMM
    with torch.cuda.device(x_slice.device):         # the eligibility query depends on the CURRENT device's CU count
        ws_bytes = lib.onebit_linear_workspace_bytes(T, Ks, shard.out_features, code)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_slice.device) if ws_bytes else None
    with torch.cuda.device(x_slice.device):
        rc = lib.onebit_matmul_partial_ws(w.data_ptr(), w.stride(0), x_slice.data_ptr(), x_slice.stride(0),
                                          shard.input_factor.data_ptr(), zp.data_ptr(),
                                          ws.data_ptr() if ws is not None else None, ws_bytes, T, Ks, shard.out_features,
                                          code, _stream_ptr(x_slice.device))
    _lib.check(rc, "onebit_matmul_partial")
    return zp


def hip_epilogue(shard: KShard, z: torch.Tensor, dtype: torch.dtype, eps: float = 1e-5, return_u: bool = False):
    """y = LayerNorm(g * round(z)) (+ bias) for complete rows through the C ABI; with `return_u` also the
    pre-LayerNorm u = fp16(fp16(z) * g) (bitnet.py:115-116), for parity tests."""
    from . import _lib
    from .bitnet import _dtype_code, _stream_ptr
    lib = _lib.load()
    T, N = z.shape
    y = torch.empty((T, N), dtype=dtype, device=z.device)
    g = shard.weight_scale.to(dtype)
    b = None if shard.bias is None else shard.bias.to(dtype)
    u = torch.empty((T, N), dtype=dtype, device=z.device) if return_u else None
    with torch.cuda.device(z.device):
        rc = lib.onebit_scale_layernorm(z.data_ptr(), g.data_ptr(), None if b is None else b.data_ptr(),
                                        y.data_ptr(), None if u is None else u.data_ptr(), T, N, _dtype_code(dtype), eps, 0,
                                        _stream_ptr(z.device))
    _lib.check(rc, "onebit_scale_layernorm")
    return (y, u) if return_u else y


def k_sharded_forward(shard: KShard, x: torch.Tensor, group=None, mode: str = "rs_ag",
                      partial_fn: Callable = hip_partial, epilogue_fn: Callable = hip_epilogue,
                      eps: float = 1e-5) -> torch.Tensor:
    """x: [T, K] full-width activations (every rank holds them, e.g. the output of the previous
    all-gather) or [T, k1-k0] already sliced.  Returns y [T, N] on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    T = x.shape[0]
    if x.shape[1] == shard.in_features:
        x_slice = x[:, shard.k0:shard.k1]          # strided view: the kernel takes a row pitch
    elif x.shape[1] == shard.k1 - shard.k0:
        x_slice = x
    else:
        raise ValueError("activation width matches neither in_features nor the shard")
    zp = partial_fn(shard, x_slice)
    if world == 1:
        return epilogue_fn(shard, zp, x.dtype, eps)
    if mode == "allreduce":
        dist.all_reduce(zp, op=dist.ReduceOp.SUM, group=group)
        return epilogue_fn(shard, zp, x.dtype, eps)
    if mode != "rs_ag":
        raise ValueError(f"unknown mode {mode}")
    # reduce_scatter over tokens (pad T to a multiple of the world size), epilogue on own rows, all_gather
    Tp = -(-T // world) * world
    if Tp != T:
        zp = torch.cat([zp, zp.new_zeros(Tp - T, zp.shape[1])], dim=0)
    rows = Tp // world
    mine = torch.empty((rows, zp.shape[1]), dtype=zp.dtype, device=zp.device)
    dist.reduce_scatter_tensor(mine, zp, op=dist.ReduceOp.SUM, group=group)
    y_mine = epilogue_fn(shard, mine, x.dtype, eps)
    y = torch.empty((Tp, zp.shape[1]), dtype=x.dtype, device=zp.device)
    dist.all_gather_into_tensor(y, y_mine, group=group)
    return y[:T]


class KShardedBitLinear(torch.nn.Module):
    """A ``BitLinearInf`` whose hidden (K) dimension is split over the ranks of ``group``: the
    module-level form of BASELINE config 4 ("decode, hidden-dim sharded across 2/4/8 GPUs with an
    RCCL all-reduce").  Every rank receives the full-width activations (they are the all-reduced
    output of the previous layer, identical everywhere), uses its K slice of them, and returns the
    complete output -- so a model whose 1-bit layers are all replaced (``shard_model_k``) computes
    the same logits on every rank with 1/world of the packed weights resident per GPU.

    For T = 1 each call moves one ``[1, N]`` fp32 vector (20-55 KB at 13B) through an all-reduce,
    280 times per token: latency-bound and slower than one GPU; the path exists to
    be measured and for models that do not fit, not as the fast decode path."""

    def __init__(self, full, rank: int, world: int, group=None, mode: str = "allreduce",
                 partial_fn: Callable = hip_partial, epilogue_fn: Callable = hip_epilogue, copy: bool = True):
        super().__init__()
        self.in_features, self.out_features = full.in_features, full.out_features
        self.shard = shard_k(full.weight.data, full.input_factor.data, full.weight_scale.data,
                             None if full.bias is None else full.bias.data, rank, world, copy=copy)
        self.group, self.mode = group, mode
        self.partial_fn, self.epilogue_fn = partial_fn, epilogue_fn

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        y = k_sharded_forward(self.shard, x.reshape(-1, x.shape[-1]), group=self.group, mode=self.mode,
                              partial_fn=self.partial_fn, epilogue_fn=self.epilogue_fn)
        return y.reshape(*lead, self.out_features)


def shard_model_k(model: torch.nn.Module, rank: int, world: int, **kw) -> torch.nn.Module:
    """Replace every ``BitLinearInf`` of ``model`` by its K shard for ``rank`` (in place)."""
    from .bitnet import BitLinearInf

    def visit(mod):
        for name, child in list(mod.named_children()):
            if isinstance(child, BitLinearInf):
                setattr(mod, name, KShardedBitLinear(child, rank, world, **kw))
            else:
                visit(child)
    visit(model)
    return model


# ---------------------------------------------------------------------------------------------------
# N-sharding (output rows split over ranks): the cheap layout on xGMI.  Rank p owns the packed rows
# N_p, g[N_p], bias[N_p] and needs the full-width activations; the only cross-rank dependency of a
# BitLinearInf is the LayerNorm over the complete output row, i.e. TWO numbers per token.  Exchange:
# one all_gather of [T, 2] fp32 (local mean and sum of squared deviations; 128 KB per rank at
# T = 16384, against 721 MB of fp32 partials for K-sharding), combined with the parallel-variance
# formula, and every rank normalises its own columns.  The output stays N-sharded (the natural input of a following K-sharded layer: gate/up
# N-sharded -> down K-sharded needs no gather in between) or is all-gathered (fp16) on request.
# ---------------------------------------------------------------------------------------------------
@dataclass
class NShard:
    weight: torch.Tensor          # int8 [n1-n0, K/8] (a row slice: contiguous view of the full matrix)
    input_factor: torch.Tensor    # [K]
    weight_scale: torch.Tensor    # [n1-n0]
    bias: Optional[torch.Tensor]  # [n1-n0]
    n0: int
    n1: int
    in_features: int
    out_features: int


def n_range(N: int, rank: int, world: int, granule: int = 16) -> Tuple[int, int]:
    """Contiguous row slice of `rank`, boundaries on multiples of `granule` (one 16-row MFMA tile)."""
    units = -(-N // granule)
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * granule, N), min(u1 * granule, N)


def shard_n(weight: torch.Tensor, input_factor: torch.Tensor, weight_scale: torch.Tensor,
            bias: Optional[torch.Tensor], rank: int, world: int) -> NShard:
    N, KB = weight.shape
    n0, n1 = n_range(N, rank, world)
    return NShard(weight[n0:n1], input_factor, weight_scale[n0:n1].contiguous(),
                  None if bias is None else bias[n0:n1].contiguous(), n0, n1, KB * 8, N)


def hip_rows_u(shard: NShard, x: torch.Tensor, prescaled: bool = False) -> torch.Tensor:
    """Pre-LayerNorm u = fp16(fp16(z) * g) of this rank's rows, [T, n1-n0], through the C ABI
    (onebit_linear_forward with ONEBIT_FLAG_SKIP_LN on the row slice).  ``prescaled``: x already holds
    fp16(x * input_factor) (ONEBIT_FLAG_PRESCALED; only where ``hip_prescaled_ok``)."""
    from . import _lib
    from .bitnet import _dtype_code, _stream_ptr
    lib = _lib.load()
    T, K = x.shape
    n = shard.n1 - shard.n0
    code = _dtype_code(x.dtype)
    u = torch.empty((T, n), dtype=x.dtype, device=x.device)
    ws_bytes = 0
    if not prescaled:
        with torch.cuda.device(x.device):
            ws_bytes = lib.onebit_linear_workspace_bytes(T, K, n, code)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    w = shard.weight
    with torch.cuda.device(x.device):
        rc = lib.onebit_linear_forward(w.data_ptr(), w.stride(0), x.data_ptr(), shard.input_factor.data_ptr(),
                                       shard.weight_scale.data_ptr(), None, u.data_ptr(), None,
                                       None if ws is None else ws.data_ptr(), ws_bytes, T, K, n, code, 0.0,
                                       _lib.FLAG_SKIP_LN | (_lib.FLAG_PRESCALED if prescaled else 0), _stream_ptr(x.device))
    _lib.check(rc, "onebit_linear_forward")
    return u


def hip_rows_u_stats(shard: NShard, x: torch.Tensor, prescaled: bool = False):
    """``hip_rows_u`` plus the local row statistics ``[T, 2]`` ({mean, sum of squared deviations}, the format of
    ``hip_row_stats``) of the rows it just produced.  Where the call takes the LDS-DMA GEMM with whole 64-row blocks
    (``onebit_linear_tile_stats_ok``) the statistics come out of the GEMM's epilogue as per-(token, 64-row block) partials
    (ONEBIT_FLAG_TILE_STATS) and ``onebit_tile_stats_combine`` reduces T * n / 64 pairs -- u is NOT read again; elsewhere
    this is ``hip_rows_u`` followed by ``hip_row_stats``."""
    from . import _lib
    from .bitnet import _dtype_code, _stream_ptr
    lib = _lib.load()
    T, K = x.shape
    n = shard.n1 - shard.n0
    code = _dtype_code(x.dtype)
    w = shard.weight
    with torch.cuda.device(x.device):
        ok = (x.dtype == torch.float16 and shard.weight_scale.dtype == torch.float16 and w.stride(-1) == 1 and w.stride(0) % 16 == 0
              and w.data_ptr() % 16 == 0 and bool(lib.onebit_linear_tile_stats_ok(T, K, n, code)))
    if not ok:
        u = hip_rows_u(shard, x, prescaled=prescaled)
        return u, hip_row_stats(u)
    u = torch.empty((T, n), dtype=x.dtype, device=x.device)
    tiles = torch.empty((T, n // 64, 2), dtype=torch.float32, device=x.device)
    ws_bytes = 0
    if not prescaled:
        with torch.cuda.device(x.device):
            ws_bytes = lib.onebit_linear_workspace_bytes(T, K, n, code)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    st = torch.empty((T, 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.onebit_linear_forward(w.data_ptr(), w.stride(0), x.data_ptr(), shard.input_factor.data_ptr(),
                                       shard.weight_scale.data_ptr(), None, u.data_ptr(), tiles.data_ptr(),
                                       None if ws is None else ws.data_ptr(), ws_bytes, T, K, n, code, 0.0,
                                       _lib.FLAG_SKIP_LN | _lib.FLAG_TILE_STATS | (_lib.FLAG_PRESCALED if prescaled else 0),
                                       _stream_ptr(x.device))
        _lib.check(rc, "onebit_linear_forward(tile stats)")
        rc = lib.onebit_tile_stats_combine(tiles.data_ptr(), st.data_ptr(), T, n, _stream_ptr(x.device))
    _lib.check(rc, "onebit_tile_stats_combine")
    return u, st


def rows_and_stats(shard: NShard, x: torch.Tensor, rows_fn: Callable, stats_fn: Callable, prescaled: bool = False):
    """(u, local row statistics) of an N-sharded layer: with the HIP callbacks one fused call (statistics out of the GEMM
    epilogue where the shape allows, ``hip_rows_u_stats``), with stand-ins (gloo tests) the two callbacks in turn."""
    if rows_fn is hip_rows_u and stats_fn is hip_row_stats:
        return hip_rows_u_stats(shard, x, prescaled=prescaled)
    u = rows_fn(shard, x, prescaled=True) if prescaled else rows_fn(shard, x)
    return u, stats_fn(u)


def hip_prescaled_ok(shard, T: int, device) -> bool:
    """True when a T-row fp16 call on this shard's matrix may take pre-scaled rows (the LDS-DMA GEMM)."""
    from . import _lib
    w = shard.weight
    n = w.shape[0]
    if shard.bias is not None or w.stride(-1) != 1 or w.stride(0) % 16 or w.data_ptr() % 16 or shard.weight_scale.dtype != torch.float16:
        return False
    with torch.cuda.device(device):
        return bool(_lib.load().onebit_linear_prescaled_ok(T, w.shape[1] * 8, n, _lib.ONEBIT_F16))


def hip_row_stats(u: torch.Tensor) -> torch.Tensor:
    """[T, 2] fp32: mean and sum of squared deviations of every row of u [T, n] (onebit_row_stats)."""
    from . import _lib
    from .bitnet import _dtype_code, _stream_ptr
    lib = _lib.load()
    T, n = u.shape
    st = torch.empty((T, 2), dtype=torch.float32, device=u.device)
    with torch.cuda.device(u.device):
        rc = lib.onebit_row_stats(u.data_ptr(), st.data_ptr(), T, n, _dtype_code(u.dtype), _stream_ptr(u.device))
    _lib.check(rc, "onebit_row_stats")
    return st


def hip_normalize(u: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    from . import _lib
    from .bitnet import _dtype_code, _stream_ptr
    lib = _lib.load()
    T, n = u.shape
    y = torch.empty_like(u)
    b = None if bias is None else bias.to(u.dtype).contiguous()
    with torch.cuda.device(u.device):
        rc = lib.onebit_normalize_rows(u.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None if b is None else b.data_ptr(),
                                       y.data_ptr(), T, n, _dtype_code(u.dtype), _stream_ptr(u.device))
    _lib.check(rc, "onebit_normalize_rows")
    return y


def _torch_row_stats(u: torch.Tensor) -> torch.Tensor:          # CPU stand-ins with the same contracts (gloo tests)
    uf = u.float()
    mean = uf.mean(dim=-1)
    return torch.stack([mean, ((uf - mean[:, None]) ** 2).sum(dim=-1)], dim=-1)


def _torch_normalize(u, mean, rstd, bias):
    y = ((u.float() - mean[:, None]) * rstd[:, None]).to(u.dtype)
    return y if bias is None else y + bias.to(u.dtype)


def n_sharded_forward(shard: NShard, x: torch.Tensor, group=None, gather: bool = False, rows_fn: Callable = hip_rows_u,
                      stats_fn: Callable = hip_row_stats, normalize_fn: Callable = hip_normalize,
                      eps: float = 1e-5) -> torch.Tensor:
    """x: [T, K] full-width activations on every rank.  Returns this rank's columns y[:, n0:n1]
    (``gather=False``) or the complete y [T, N] (``gather=True``, needs equal slices)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    u, st = rows_and_stats(shard, x, rows_fn, stats_fn)                # [T, n_p] in x.dtype; [T, 2]: local mean, local M2
    N = shard.out_features
    n_p = float(shard.n1 - shard.n0)
    if world > 1:
        # parallel variance (Chan et al.): N*mean = sum n_p*mean_p;  M2 = sum M2_p + sum n_p*(mean_p - mean)^2
        all_st = torch.empty((world,) + tuple(st.shape), dtype=st.dtype, device=st.device)
        counts = torch.tensor([n_range(N, r, world)[1] - n_range(N, r, world)[0] for r in range(world)],
                              dtype=torch.float32, device=st.device).view(world, 1)
        dist.all_gather_into_tensor(all_st.view(world * st.shape[0], 2), st.contiguous(), group=group)
        mean = (all_st[:, :, 0] * counts).sum(dim=0) / N
        m2 = all_st[:, :, 1].sum(dim=0) + (counts * (all_st[:, :, 0] - mean[None]) ** 2).sum(dim=0)
    else:
        mean, m2 = st[:, 0].contiguous(), st[:, 1].contiguous()
        del n_p
    rstd = torch.rsqrt(m2 / N + eps)
    y = normalize_fn(u, mean.contiguous(), rstd.contiguous(), shard.bias)
    if not gather or world == 1:
        return y
    n = shard.n1 - shard.n0
    if n * world != N:
        raise ValueError("gather=True needs equal row slices")
    T = y.shape[0]
    parts = torch.empty((world * T, n), dtype=y.dtype, device=y.device)      # rank-major concatenation
    dist.all_gather_into_tensor(parts, y.contiguous(), group=group)
    return parts.view(world, T, n).permute(1, 0, 2).reshape(T, N)


# ---------------------------------------------------------------------------------------------------
# Static-shape greedy decode: one token per step with the token id and the position ON THE DEVICE and every tensor
# shape fixed, so that a whole step of a K-sharded model -- 7 partial GEMVs + 7 all-reduces per layer -- is ONE HIP
# graph (RCCL collectives are stream-ordered and capturable).  BASELINE config 4 measures the exchange; run eagerly the
# same step is ~600 Python-issued launches per token and measures the host instead.  Works on any model of this package
# (sharded or not: with world 1 it is the single-GPU module arithmetic under a graph).
# Attention runs over the WHOLE preallocated cache with positions beyond the current one masked (additive
# finfo.min, the reference's mask form, modeling_bitllama.py:1256-1262): their probabilities are exactly 0.
# ---------------------------------------------------------------------------------------------------
class StaticShapeDecoder:
    def __init__(self, model: torch.nn.Module, max_len: int, use_graph: bool = True):
        from .llama import _rotate_half
        self._rot = _rotate_half
        self.model, self.cfg = model, model.config
        p = model.lm_head.weight
        self.dev, self.dtype = p.device, p.dtype
        self.max_len = int(max_len)
        self.cache = model.new_cache(1, self.max_len)
        self.cos, self.sin = model._rope_tables(self.dev, self.dtype, self.max_len)
        self.tok = torch.zeros((1, 1), dtype=torch.long, device=self.dev)
        self.pos = torch.zeros(1, dtype=torch.long, device=self.dev)
        self.out_tokens = torch.zeros(self.max_len, dtype=torch.long, device=self.dev)
        self._ar = torch.arange(self.max_len, device=self.dev)
        self.use_graph = bool(use_graph) and p.is_cuda
        self.graph = None
        self.steps = 0

    @torch.no_grad()
    def prime(self, prompt: torch.Tensor) -> int:
        """Prompt through the module path into the static cache; arms token / position.  Returns the first new token."""
        S = prompt.shape[1]
        if S + 1 > self.max_len:
            raise ValueError("prompt longer than max_len")
        self.cache.length = 0
        logits = self.model(prompt.to(self.dev), self.cache)
        self.tok.copy_(logits[:, -1].argmax(-1, keepdim=True))
        self.pos.fill_(S)
        self.steps = S
        return int(self.tok.item())

    @torch.no_grad()
    def _step(self):
        m, cfg = self.model.model, self.cfg
        H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        pos = self.pos
        h = m.embed_tokens(self.tok)                                        # [1, 1, hidden]
        c = self.cos.index_select(0, pos)[None, None]                       # [1, 1, 1, D]
        s = self.sin.index_select(0, pos)[None, None]
        mask = torch.where(self._ar > pos, torch.finfo(self.dtype).min, 0.0).to(self.dtype)[None, None, None]   # [1,1,1,max_len]
        for layer, (kc, vc) in zip(m.layers, self.cache.layers):
            att = layer.self_attn
            x = layer.input_layernorm(h)
            q = att.q_proj(x).view(1, 1, H, D).transpose(1, 2)
            k = att.k_proj(x).view(1, 1, Hkv, D).transpose(1, 2)
            v = att.v_proj(x).view(1, 1, Hkv, D).transpose(1, 2)
            q = (q * c) + (self._rot(q) * s)
            k = (k * c) + (self._rot(k) * s)
            kc.index_copy_(2, pos, k)
            vc.index_copy_(2, pos, v)
            keys, vals = kc, vc
            if Hkv != H:
                keys, vals = keys.repeat_interleave(H // Hkv, dim=1), vals.repeat_interleave(H // Hkv, dim=1)
            w = torch.matmul(q, keys.transpose(2, 3)) / math.sqrt(D) + mask
            w = torch.nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
            o = torch.matmul(w, vals).transpose(1, 2).reshape(1, 1, H * D)
            h = h + att.o_proj(o)
            h = h + layer.mlp(layer.post_attention_layernorm(h))
        logits = self.model.lm_head(m.norm(h)).float()
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        self.out_tokens.index_copy_(0, pos, nxt.view(1))
        self.tok.copy_(nxt)
        self.pos.add_(1)
        self._logits = logits

    def step(self):
        if self.steps >= self.max_len:
            raise RuntimeError("StaticShapeDecoder: KV cache full")
        if not self.use_graph:
            self._step()
        else:
            if self.graph is None:
                # warm up on a side stream (communicators, allocator pools), restoring the state it advanced
                tok0, pos0 = self.tok.clone(), self.pos.clone()
                side = torch.cuda.Stream(self.dev)
                side.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(side):
                    self._step()
                torch.cuda.current_stream(self.dev).wait_stream(side)
                self.tok.copy_(tok0); self.pos.copy_(pos0)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step()
                self.tok.copy_(tok0); self.pos.copy_(pos0)                  # capture does not execute; state as before
                self.graph = g
            self.graph.replay()
        self.steps += 1


# ---------------------------------------------------------------------------------------------------
# BASELINE config 4, fused (round 5): K-sharded single-stream decode as NATIVE segments + 4 collectives per layer.
#
# ``StaticShapeDecoder`` above replays the sharded MODULES (~15 launches per BitLinearInf-terminated op group, one
# all-reduce per BitLinearInf: 280 collectives and ~600 launches per 13B token; 101 tok/s at N = 1 against 620 for the fused
# single-GPU engine).  Here the step is ``onebit_decode_step_ksharded`` (include/onebit.h): per layer four segments of
# native kernels -- the decode GEMV in its fp32-partial form on the rank's K slice (q|k|v and gate|up as ONE launch and ONE
# fp32 buffer each), one-workgroup row kernels for the replicated glue, the decode attention kernel (the consumers of a reduced sum
# round and scale it themselves) -- and between them ONE all-reduce of the named buffer: 4 collectives and 8 launches per layer,
# all of it under one HIP graph.
#
#   z_qkv [NQ + 2 NK] -> all_reduce -> z_o [H] -> all_reduce -> z_gu [2 I] -> all_reduce -> z_down [H] -> all_reduce
#
# Slices are windows into the FULL packed matrices (no copy; boundaries on multiples of 128 columns = 16 bytes of a packed
# row, the alignment of the decode kernels' 16-byte weight loads).  Everything that is not a K-sliced product is computed by
# every rank from identical (all-reduced) inputs.  The segment backend is pluggable so that the exchange protocol -- which
# buffer is reduced when, which columns a rank multiplies -- runs under gloo on CPU with a torch statement of the five
# segments (tests/test_sharded_cpu.py); the product backend is the C ABI and nothing else.
# ---------------------------------------------------------------------------------------------------
class _HipSegments:
    """The five segments through ``onebit_decode_step_ksharded`` (no CPU fallback)."""

    def __init__(self, dec: "FusedKShardedDecoder"):
        import ctypes
        from . import _lib
        from .engine import _KState, _Model, _model_struct, fp16_view
        if not dec.dev.type == "cuda":
            raise RuntimeError("FusedKShardedDecoder needs the model on a ROCm GPU (no CPU fallback)")
        self._ct, self._lib_mod = ctypes, _lib
        self.lib = _lib.load()
        self.dec = dec
        self._model, self._keep = _model_struct(dec.model, dec.cache.layers, dec.max_len, krange=dec.kr)
        b = dec.buf
        self.lib.onebit_decode_stats_floats.restype = ctypes.c_size_t
        self.lib.onebit_decode_stats_floats.argtypes = [ctypes.POINTER(_Model)]
        self._tile_stats = torch.zeros(max(int(self.lib.onebit_decode_stats_floats(ctypes.byref(self._model))), 1),
                                       dtype=torch.float32, device=dec.dev)
        p = lambda t: t.data_ptr()
        self._state = _KState(ctypes.sizeof(_KState), p(dec.token), p(dec.pos), p(dec.out_tokens), dec.max_len,
                              p(b["hres0"]), p(b["hres1"]), p(b["x"]), p(b["u_q"]), p(b["u_k"]), p(b["u_v"]), p(b["attn_out"]),
                              p(b["u_gate"]), p(b["u_up"]), p(b["act"]), p(b["u_down"]),
                              p(dec.z_qkv), p(dec.z_o), p(dec.z_gu), p(dec.z_down),
                              p(b["logits"]), p(b["part_val"]), p(b["part_idx"]), p(self._tile_stats),
                              dec.kr(dec.cfg.hidden_size)[0], dec.kr(dec.cfg.num_attention_heads * dec.cfg.head_dim)[0],
                              dec.kr(dec.cfg.intermediate_size)[0], 0, 0, None)
        if dec.attn_chunk:
            # key-block attention: ONE graph serves every context, so the split count covers max_len (splits past a step's
            # context leave at once)
            splits = -(-dec.max_len // dec.attn_chunk)
            if splits > 64:
                raise ValueError(f"attn_chunk {dec.attn_chunk} gives {splits} > 64 attention splits at max_len {dec.max_len}")
            self.lib.onebit_attention_decode_scratch_bytes.restype = ctypes.c_size_t
            nb = int(self.lib.onebit_attention_decode_scratch_bytes(1, dec.cfg.num_attention_heads, splits))
            self._attn_scratch = torch.zeros(max(nb, 16), dtype=torch.uint8, device=dec.dev)
            self._state.attn_chunk, self._state.attn_splits = dec.attn_chunk, splits
            self._state.attn_scratch = self._attn_scratch.data_ptr()
        self.lib.onebit_decode_step_ksharded.restype = ctypes.c_int
        self.lib.onebit_decode_step_ksharded.argtypes = [ctypes.POINTER(_Model), ctypes.POINTER(_KState), ctypes.c_int32, ctypes.c_int32,
                                                         ctypes.c_void_p]

    def segment(self, layer: int, seg: int):
        dev = self.dec.dev
        with torch.cuda.device(dev):
            rc = self.lib.onebit_decode_step_ksharded(self._ct.byref(self._model), self._ct.byref(self._state), layer, seg,
                                                      torch.cuda.current_stream(dev).cuda_stream)
        self._lib_mod.check(rc, "onebit_decode_step_ksharded")


class FusedKShardedDecoder:
    """Greedy single-stream decode of ``model`` with every 1-bit projection's K dimension sharded over ``world`` ranks
    (this process is ``rank``); ``model`` is the complete fp16 checkpoint on this rank's device -- the rank READS only its
    column window of every packed matrix (1/world of the weight bytes per token).  ``group``: the process group of the K
    shards (default: the world group when ``world`` > 1).  ``reduce_fn(t)``: replaces the all-reduce (tests drive several
    ranks in one process).  ``backend(dec)``: an object with ``segment(layer, seg)`` (default: the C ABI).  ``granule``:
    slice boundaries in columns (the C ABI needs 128; CPU stand-ins of the segments may use 32 on toy widths).
    ``attn_chunk``: cached tokens per attention workgroup of the key-block form (``onebit_kshard_state_t.attn_chunk``); None = that
    form with 256 (512, ... : at most 64 workgroups per head) tokens per workgroup when ``max_len`` > ``long_context_from``, else the
    one-workgroup-per-head launch (one launch fewer per layer; its LDS holds a score per cached token); 0 = always the latter.
    13B shapes, N = 1 (tools/kshard_ctx_probe.py): 2.03 ms / token either way at 16 cached tokens, 2.07 vs 2.11 at 128,
    2.85 vs 2.21 at 512, 5.05 vs 2.41 at 2000."""

    SEGMENTS = (0, 1, 2, 3)          # ONEBIT_KSEG_QKV, _ATTN_O, _GATE_UP, _DOWN; 4 = _HEAD once per token

    def __init__(self, model: torch.nn.Module, rank: int, world: int, max_len: int, group=None, use_graph: bool = True,
                 reduce_fn: Optional[Callable] = None, backend: Optional[Callable] = None, granule: int = 128,
                 attn_chunk: Optional[int] = None, long_context_from: int = 256):
        from .llama import KVCache
        cfg = model.config
        p = model.lm_head.weight
        if backend is None:
            from .engine import fp16_view
            model = fp16_view(model)
            p = model.lm_head.weight
        self.model, self.cfg, self.dev = model, cfg, p.device
        self.rank, self.world, self.group = int(rank), int(world), group
        if not 0 <= self.rank < self.world:
            raise ValueError("rank outside the world")
        self.max_len = int(max_len)
        if self.max_len > cfg.max_position_embeddings:
            raise ValueError("max_len exceeds max_position_embeddings")
        H, I, D = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
        NQ, NK = cfg.num_attention_heads * D, cfg.num_key_value_heads * D
        if attn_chunk is None:
            attn_chunk = 0
            if self.max_len > long_context_from and D >= 16 and D & (D - 1) == 0:
                attn_chunk = 256
                while -(-self.max_len // attn_chunk) > 64:
                    attn_chunk *= 2
        if attn_chunk < 0 or attn_chunk % 64:
            raise ValueError("attn_chunk must be a non-negative multiple of 64")
        self.attn_chunk = int(attn_chunk)
        if backend is None and granule % 128:
            raise ValueError("the HIP segments need slice boundaries on multiples of 128 columns (16-byte weight loads)")
        for K in (H, NQ, I):
            if K % granule or K // granule < self.world:
                raise ValueError(f"in_features={K} cannot be split into {self.world} slices of whole {granule}-column granules")
        self.kr = lambda K: k_range(K, self.rank, self.world, granule=granule)
        dev, f16 = self.dev, p.dtype
        z = lambda n, dt=f16: torch.zeros(n, dtype=dt, device=dev)
        self.cache = KVCache(cfg, 1, self.max_len, dev, f16)
        self.token, self.pos = z(1, torch.int32), z(1, torch.int32)
        self.out_tokens = z(self.max_len, torch.int32)
        self.buf = dict(hres0=z(H), hres1=z(H), x=z(H), u_q=z(NQ), u_k=z(NK), u_v=z(NK), attn_out=z(NQ), u_gate=z(I), u_up=z(I),
                        act=z(I), u_down=z(H), logits=z(cfg.vocab_size), part_val=z(1024, torch.float32), part_idx=z(1024, torch.int32))
        # the four exchanged buffers: fp32 partial sums out of a segment, complete sums into the next
        self.z_qkv, self.z_o = z(NQ + 2 * NK, torch.float32), z(H, torch.float32)
        self.z_gu, self.z_down = z(2 * I, torch.float32), z(H, torch.float32)
        self.collectives_per_token = 4 * cfg.num_hidden_layers if self.world > 1 else 0
        self._reduce_fn = reduce_fn
        self.backend = _HipSegments(self) if backend is None else backend(self)
        self.use_graph = bool(use_graph) and p.is_cuda
        self.graph = None
        self.steps = 0
        self.first_token = None

    # ---- one token -------------------------------------------------------------------------------
    def _reduce(self, t: torch.Tensor):
        if self._reduce_fn is not None:
            self._reduce_fn(t)
        elif self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _segments(self):
        """Launch segment after segment; yields the buffer the caller has to all-reduce before the next one."""
        for l in range(self.cfg.num_hidden_layers):
            for seg, buf in zip(self.SEGMENTS, (self.z_qkv, self.z_o, self.z_gu, self.z_down)):
                self.backend.segment(l, seg)
                yield buf
        self.backend.segment(0, 4)

    def _step(self):
        for buf in self._segments():
            self._reduce(buf)

    @torch.no_grad()
    def prime(self, prompt: torch.Tensor) -> int:
        """The prompt through the (replicated) module path into the KV cache; arms token / position."""
        if prompt.dim() != 2 or prompt.shape[0] != 1:
            raise ValueError("FusedKShardedDecoder is batch 1: prompt must be [1, S]")
        S = prompt.shape[1]
        if S + 1 > self.max_len:
            raise ValueError("prompt longer than max_len")
        self.cache.length = 0
        logits = self.model(prompt.to(self.dev), self.cache)
        self.token.copy_(logits[0, -1].argmax().to(torch.int32).reshape(1))
        self.pos.fill_(S)
        self.steps = self._prompt_len = S
        self.first_token = int(self.token.item())
        return self.first_token

    def set_state(self, token: int, pos: int):
        self.token.fill_(int(token)); self.pos.fill_(int(pos)); self.steps = int(pos)

    @torch.no_grad()
    def capture(self) -> bool:
        """Warm-up + capture of one step (kernels AND collectives) into a HIP graph.  Returns False -- the decoder then keeps
        stepping eagerly -- if the capture raised: with world > 1 the CALLER must agree on the outcome across ranks before anybody
        replays (an all-reduce of the flag: bench.py), or one rank replays collectives the others issue eagerly."""
        if self.graph is not None:
            return True
        tok0, pos0 = self.token.clone(), self.pos.clone()
        try:
            side = torch.cuda.Stream(self.dev)           # warm-up off the capture: function attributes, communicators
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._step()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            self.token.copy_(tok0); self.pos.copy_(pos0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step()
            self.graph = g
            return True
        except Exception as e:                            # (a failed capture leaves the stream usable: torch ends it on the way out)
            self.capture_error = "%s: %s" % (type(e).__name__, e)
            self.graph = None
            self.use_graph = False
            return False
        finally:
            self.token.copy_(tok0); self.pos.copy_(pos0)

    @torch.no_grad()
    def step(self):
        if self.steps >= self.max_len:
            raise RuntimeError("FusedKShardedDecoder: KV cache full")
        if self.use_graph and self.graph is None:
            self.capture()
        if self.use_graph and self.graph is not None:
            self.graph.replay()
        else:
            self._step()
        self.steps += 1

    def logits(self) -> torch.Tensor:
        return self.buf["logits"].float()

    @torch.no_grad()
    def generate(self, prompt: torch.Tensor, max_new_tokens: int) -> List[int]:
        first = self.prime(prompt)
        for _ in range(max(max_new_tokens - 1, 0)):
            self.step()
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        n = max(max_new_tokens - 1, 0)
        return ([first] if max_new_tokens > 0 else []) + self.out_tokens[self._prompt_len:self._prompt_len + n].tolist()


def lockstep_step(decoders) -> None:
    """One token of several ``FusedKShardedDecoder`` ranks living in ONE process (tests): every rank's segment, then the
    sum of the ranks' partial buffers written back to all of them -- what the all-reduce does, in rank order."""
    gens = [d._segments() for d in decoders]
    while True:
        bufs = []
        for g in gens:
            try:
                bufs.append(next(g))
            except StopIteration:
                pass
        if not bufs:
            break
        total = torch.stack(bufs).sum(0)
        for b in bufs:
            b.copy_(total)
    for d in decoders:
        d.steps += 1
