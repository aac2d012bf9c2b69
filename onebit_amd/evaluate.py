"""Evaluation callers of the 1-bit model -- SURVEY.md section 8(f) rank 3.

Host-side restatements of the two loops the reference drives its inference model with:

* ``perplexity``  -- the PPL loop of ``evaluation/lm_eval.py:93-128``: the token stream is cut into
  ``numel // seqlen`` windows, each window is one prefill, the loss is ``nn.CrossEntropyLoss`` over
  the shifted logits IN THE LOGITS' DTYPE (fp16 for an fp16 model), and -- as the reference does
  -- the mean over ``seqlen - 1`` predictions is scaled by ``seqlen`` (``:121``), so the reported
  number is ``exp(sum(nll) / (nsamples * seqlen))`` (``:126``).  ``limit`` reproduces the
  reference's early exit (``:123``: it stops after window index ``limit``, i.e. ``limit + 1``
  windows, and still divides by ``nsamples``).
* ``loglikelihood_tokens`` -- ``BaseLM._loglikelihood_tokens``
  (``evaluation/lm_eval/models_utils.py:257-438``): requests sorted by descending total length,
  chunks of ``batch_size``, inputs ``(context + continuation)[-(max_length + 1):][:-1]`` right-padded
  with token 0 to the first (longest) request of the chunk, one batched prefill WITHOUT an attention
  mask (causal attention keeps the padding harmless), ``log_softmax`` over the vocabulary, the
  continuation's log-probabilities summed, plus the "greedy decoding would have produced exactly
  this continuation" flag.  Results are returned in the callers' order.

Multi-GPU: the reference spreads the decoder LAYERS over the visible GPUs by free memory
(``evaluation/lm_eval/parallel_utils.py:12-36,89-130``, driven by ``nvidia-smi``) because a dense fp16
13B model does not fit its cards.  The packed model is 0.8-1.6 GB: on MI355X every rank keeps all of it
and the WORK is sharded instead -- ``perplexity`` windows / ``loglikelihood_tokens`` chunks are dealt
round-robin to the ranks of a ``torch.distributed`` group (``rank`` / ``world`` / ``group`` arguments;
RCCL on GPUs, gloo in the CPU tests) and the per-window / per-request results are all-gathered, so every
rank returns exactly what a single process returns (same values, same summation order).

Both run prefill-shaped work: T = B * S tokens per 1-bit layer, i.e. the MFMA GEMM of
``onebit_linear_forward`` with ragged, padded shapes.  The model is called as the reference calls
its own (``model(inps)`` -> logits ``[B, S, vocab]``); nothing here touches the oracle.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn
import torch.nn.functional as F


def _model_logits(model, inps: torch.Tensor) -> torch.Tensor:
    """``LMClass._model_call`` (evaluation/lm_eval/LMClass.py:78-87): logits of one batched prefill."""
    out = model(inps)
    return out["logits"] if isinstance(out, dict) else out


@torch.no_grad()
def perplexity(model, token_ids: torch.Tensor, seqlen: int, limit: int = -1,
               logits_dtype: Optional[torch.dtype] = None, rank: int = 0, world: int = 1, group=None) -> float:
    """PPL of ``token_ids`` ([1, N] long) in windows of ``seqlen`` (lm_eval.py:93-128).

    ``logits_dtype``: dtype the loss is computed in.  The reference computes
    ``lm_head(hidden_states)`` itself (``:104-105``), so its logits -- and hence the loss -- are
    in the parameter dtype; pass ``torch.float16`` to reproduce that with a model whose ``forward``
    returns fp32 logits (``OneBitLlamaForCausalLM`` follows ``BitLlamaForCausalLMInf.forward``,
    which upcasts).  ``None`` keeps whatever the model returns.
    """
    if token_ids.dim() != 2 or token_ids.shape[0] != 1:
        raise ValueError("token_ids must be [1, N]")
    nsamples = token_ids.numel() // seqlen
    if nsamples == 0:
        raise ValueError("fewer tokens than one window")
    dev = next(model.parameters()).device
    loss_fct = nn.CrossEntropyLoss()
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank / world")
    nwin = nsamples if limit < 0 or limit >= nsamples else limit + 1       # (:123: stops after window index `limit`)
    mine = torch.zeros(nwin, dtype=torch.float32, device=dev)
    for i in range(rank, nwin, world):
        batch = token_ids[:, i * seqlen:(i + 1) * seqlen].to(dev)
        logits = _model_logits(model, batch)
        if logits_dtype is not None:
            logits = logits.to(logits_dtype)
        shift_logits = logits[:, :-1, :]
        shift_labels = batch[:, 1:]
        loss = loss_fct(shift_logits.reshape(-1, shift_logits.size(-1)), shift_labels.reshape(-1))
        mine[i] = loss.float() * seqlen
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(mine, op=dist.ReduceOp.SUM, group=group)            # disjoint supports: a gather by addition of zeros
    return float(torch.exp(mine.sum() / (nsamples * seqlen)).item())


@torch.no_grad()
def _ragged_scorer(model, n_slots: int, max_len: int, any_dtype: bool = False):
    """A ``MixedStep`` over ``n_slots`` fresh KV-cache slots for the ragged route of ``loglikelihood_tokens``; None when the
    native step does not take the model (CPU, another model class, head_dim other than 64 / 128)."""
    try:
        from .engine import MixedStep, fp16_view
        from .llama import KVCache, OneBitLlamaForCausalLM
        if not isinstance(model, OneBitLlamaForCausalLM) or not model.lm_head.weight.is_cuda:
            return None
        if model.lm_head.weight.dtype != torch.float16 and not any_dtype:      # (an fp32 checkpoint keeps its fp32 logits unless asked)
            return None
        m16 = fp16_view(model)
        cache = KVCache(m16.config, n_slots, max_len, m16.lm_head.weight.device, torch.float16)
        return MixedStep(m16, cache.layers, n_slots, max_len, max_rows=64, keep_logits=True)
    except (ValueError, ImportError):
        return None


@torch.no_grad()
def loglikelihood_tokens(model, requests: Sequence[Tuple[Sequence[int], Sequence[int]]], batch_size: int,
                         max_length: int, vocab_size: Optional[int] = None, rank: int = 0, world: int = 1,
                         group=None, ragged: Optional[bool] = None, max_rows: int = 16384) -> List[Tuple[float, bool]]:
    """``[(sum log p(continuation | context), is_greedy)]`` for ``requests = [(context_ids,
    continuation_ids)]`` (models_utils.py:257-438).  ``vocab_size`` is the reference's
    ``[:, :, :self.vocab_size]`` slice (tokenizer vocabulary; default: all logits).
    ``world > 1``: chunk ``c`` of the sorted requests is evaluated by rank ``c % world``, the results are
    exchanged with ``all_gather_object``; every rank returns the complete list.
    ``ragged`` (round 6; None = wherever the native step takes the model): a chunk runs as ONE ``onebit_mixed_step`` over the
    requests' REAL token rows -- no right-padding to the longest request (the reference's batch of 32 requests of 44..390
    tokens is 42 % padding), lm_head on the continuation rows only -- at most ``max_rows`` rows per call.  Causal attention
    makes the padding irrelevant to the scored rows, so both routes score the same numbers up to fp16 GEMM-route noise."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank / world")
    dev = next(model.parameters()).device
    scorer = None
    if ragged is None or ragged:
        longest = max((min(len(c) + len(t), max_length + 1) - 1 for c, t in requests), default=0)
        scorer = _ragged_scorer(model, max(1, min(batch_size, len(requests))), max(longest, 1), any_dtype=bool(ragged)) if longest >= 1 else None
        if ragged and scorer is None:
            raise RuntimeError("loglikelihood_tokens(ragged=True): the native mixed step does not take this model / device")
    # Reorderer (models_utils.py:544-568): requests whose concatenated tokens are identical form ONE
    # group, evaluated once with the first member's (context, continuation) split -- the split is
    # not part of the key, a quirk kept here -- and groups are ordered by (-length, tokens).
    groups = {}
    for i, (ctx, cont) in enumerate(requests):
        toks = tuple(ctx) + tuple(cont)
        groups.setdefault((-len(toks), toks), []).append(i)
    order = sorted(groups)
    res: List[Optional[Tuple[float, bool]]] = [None] * len(requests)
    for ci, c0 in enumerate(range(0, len(order), batch_size)):
        if ci % world != rank:
            continue
        chunk = order[c0:c0 + batch_size]
        inps, inplens, conts = [], [], []
        padding_length = None
        for key in chunk:
            ctx, cont = (list(t) for t in requests[groups[key][0]])
            if not ctx or not cont or len(cont) > max_length:
                raise ValueError("empty context/continuation or continuation longer than max_length")
            inp = torch.tensor((ctx + cont)[-(max_length + 1):][:-1], dtype=torch.long)
            inplen = inp.shape[0]
            padding_length = padding_length if padding_length is not None else inplen
            inps.append(torch.cat([inp, torch.zeros(padding_length - inplen, dtype=torch.long)]).unsqueeze(0))
            inplens.append(inplen)
            conts.append(cont)
        if scorer is not None:
            # ragged route: items = (slot, 0, the request's input tokens, its continuation length); sub-batches of <= max_rows rows
            lg, items, rows = [], [], 0
            for b, (inp, n, c) in enumerate(zip(inps, inplens, conts)):
                if items and rows + n > max_rows:
                    scorer.launch(items)
                    lg.append(scorer.logits[:sum(it[3] for it in items)].float())
                    items, rows = [], 0
                items.append((len(items), 0, inp[0, :n].tolist(), len(c)))
                rows += n
            scorer.launch(items)
            lg.append(scorer.logits[:sum(it[3] for it in items)].float())
            cont_rows = lg[0] if len(lg) == 1 else torch.cat(lg, dim=0)        # [sum(contlen), vocab], fp32 of the fp16 lm_head output
        else:
            batched = torch.cat(inps, dim=0).to(dev)
            logits = _model_logits(model, batched)                              # [B, S, vocab] on the device
        # Score ON THE DEVICE (round 6).  The reference ships log_softmax of the whole [B, S, vocab] tensor to the host
        # (models_utils.py:331-334: 1.6 GB for a 32 x 389 batch of a 32000-word vocabulary) to read sum(contlen) numbers from it.
        # log_softmax is row-wise, so it is taken over the continuation rows [inplen - contlen, inplen) only -- the same values --,
        # the continuation's log-probabilities are gathered and the greedy flags formed there; the host receives sum(contlen)
        # floats + flags and adds each request's log-probabilities in the reference's order (fp32 on the CPU, :352).
        cont_t = torch.tensor([t for c in conts for t in c], dtype=torch.long, device=dev)
        if scorer is None:
            b_idx = torch.tensor([b for b, (n, c) in enumerate(zip(inplens, conts)) for _ in c], dtype=torch.long, device=dev)
            p_idx = torch.tensor([n - len(c) + j for n, c in zip(inplens, conts) for j in range(len(c))], dtype=torch.long, device=dev)
            cont_rows = logits[b_idx, p_idx]
        lsm = F.log_softmax(cont_rows, dim=-1)                                   # [sum(contlen), vocab]
        if vocab_size is not None:
            lsm = lsm[:, :vocab_size]
        greedy_tok = (lsm.argmax(dim=-1) == cont_t).cpu()
        lp_all = torch.gather(lsm, 1, cont_t.unsqueeze(-1)).squeeze(-1).cpu()
        o = 0
        for key, cont in zip(chunk, conts):
            contlen = len(cont)
            lp, is_greedy = lp_all[o:o + contlen], bool(greedy_tok[o:o + contlen].all())
            o += contlen
            for i in groups[key]:
                res[i] = (float(lp.sum()), is_greedy)
    if world > 1:
        import torch.distributed as dist
        parts: List[Optional[list]] = [None] * world
        dist.all_gather_object(parts, res, group=group)
        for part in parts:
            for i, v in enumerate(part):            # each request was evaluated by exactly one rank
                if v is not None:
                    res[i] = v
    return res  # type: ignore[return-value]
