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

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["KShard", "shard_k", "k_sharded_forward", "hip_partial", "hip_epilogue", "KShardedBitLinear",
           "shard_model_k"]


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
    with torch.cuda.device(x_slice.device):
        rc = lib.onebit_matmul_partial(w.data_ptr(), w.stride(0), x_slice.data_ptr(), x_slice.stride(0),
                                       shard.input_factor.data_ptr(), zp.data_ptr(), T, Ks, shard.out_features,
                                       _dtype_code(x_slice.dtype), _stream_ptr(x_slice.device))
    _lib.check(rc, "onebit_matmul_partial")
    return zp


def hip_epilogue(shard: KShard, z: torch.Tensor, dtype: torch.dtype, eps: float = 1e-5) -> torch.Tensor:
    """y = LayerNorm(g * round(z)) (+ bias) for complete rows through the C ABI."""
    from . import _lib
    from .bitnet import _dtype_code, _stream_ptr
    lib = _lib.load()
    T, N = z.shape
    y = torch.empty((T, N), dtype=dtype, device=z.device)
    g = shard.weight_scale.to(dtype)
    b = None if shard.bias is None else shard.bias.to(dtype)
    with torch.cuda.device(z.device):
        rc = lib.onebit_scale_layernorm(z.data_ptr(), g.data_ptr(), None if b is None else b.data_ptr(),
                                        y.data_ptr(), None, T, N, _dtype_code(dtype), eps, 0, _stream_ptr(z.device))
    _lib.check(rc, "onebit_scale_layernorm")
    return y


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
