"""Continuous batching for the 1-bit model -- SURVEY.md section 8(f) rank 4 / BASELINE config 5
("mixed prefill + decode continuous batch").  The reference has no counterpart: its ``generate`` is
one sequence batch per call.

What matters on this hot path: every scheduled token of a step -- whole prompts of newly admitted
requests AND the single next token of every running request -- goes through each 1-bit projection
in ONE ``BitLinearInf`` call on the concatenated ``[T, hidden]`` activations, so the packed weights
(the dominant HBM bytes at small T) are streamed once per step for all sequences; only attention
is per request (its own KV-cache slot, its own positions).

``Scheduler`` is pure Python (slot admission, step plans, completion) and is unit-tested on the
CPU; ``ContinuousBatcher`` executes plans on the GPU with the model's own modules and the exact
op order of ``LlamaAttentionInf.forward`` (modeling_bitllama.py:487-585), so a request's tokens
equal ``model.generate`` on that request alone up to fp16 accumulation order.

Multi-GPU: requests are independent -- shard them across ranks (one ``ContinuousBatcher`` per
GPU, weights replicated: 0.8-1.6 GB); there is no data-path collective to add.
"""
from __future__ import annotations

import math
import time
from collections import deque
from dataclasses import dataclass, field
from typing import Deque, Dict, List, Optional, Tuple

import torch
from torch import nn

from .llama import OneBitLlamaForCausalLM, _rotate_half


@dataclass
class Request:
    rid: int
    prompt: List[int]
    max_new_tokens: int
    out: List[int] = field(default_factory=list)
    slot: Optional[int] = None
    pos: int = 0                      # tokens of this request already in its KV-cache slot
    t_add: float = 0.0                # wall clock (time.perf_counter): submitted / first token produced / finished
    t_first: float = 0.0
    t_done: float = 0.0

    @property
    def done(self) -> bool:
        return len(self.out) >= self.max_new_tokens


@dataclass
class Item:                           # one request's share of a step
    req: Request
    tokens: List[int]
    start: int                        # position of tokens[0]


class Scheduler:
    """FIFO admission into ``max_batch`` KV-cache slots; a step schedules the whole prompt of every
    newly admitted request (bounded by ``max_step_tokens``) plus one token per running request.

    ``prefill_chunk``: chunked prefill -- a prompt enters the cache at most that many tokens per step, so
    a long prompt no longer holds every decoding request's next token back for the length of its whole
    prefill (a 2048-token prompt is ~30 ms of GEMM at 7B; in chunks of 256 the running requests see
    ~4 ms steps instead).  A request samples its first token in the step that takes its last chunk."""

    def __init__(self, max_batch: int, max_len: int, max_step_tokens: Optional[int] = None,
                 prefill_chunk: Optional[int] = None):
        if max_batch <= 0 or max_len <= 0:
            raise ValueError("max_batch and max_len must be positive")
        if prefill_chunk is not None and prefill_chunk <= 0:
            raise ValueError("prefill_chunk must be positive")
        if max_step_tokens is not None:
            if max_step_tokens < 1:
                raise ValueError("max_step_tokens must be >= 1")
            # decode tokens come first in a step; with chunked prefill the budget must leave room for at least one
            # prompt token once every slot is decoding, or a half-entered prompt holds its KV slot and starves
            if prefill_chunk is not None and max_step_tokens <= max_batch:
                raise ValueError("with prefill_chunk, max_step_tokens must exceed max_batch "
                                 "(one token per decoding slot plus at least one prompt token per step)")
        self.max_batch, self.max_len = max_batch, max_len
        self.max_step_tokens = max_step_tokens
        self.prefill_chunk = prefill_chunk
        self.waiting: Deque[Request] = deque()
        self.running: List[Request] = []
        self.finished: Dict[int, Request] = {}
        self._free = list(range(max_batch - 1, -1, -1))
        self._next = 0

    def add(self, prompt: List[int], max_new_tokens: int) -> int:
        if not prompt or max_new_tokens <= 0:
            raise ValueError("empty prompt or max_new_tokens <= 0")
        if len(prompt) + max_new_tokens - 1 > self.max_len:
            raise ValueError("request does not fit max_len")
        if self.max_step_tokens is not None and self.prefill_chunk is None and len(prompt) > self.max_step_tokens:
            raise ValueError("prompt longer than max_step_tokens")
        r = Request(self._next, list(prompt), max_new_tokens, t_add=time.perf_counter())
        self._next += 1
        self.waiting.append(r)
        return r.rid

    @property
    def idle(self) -> bool:
        return not self.waiting and not self.running

    def _chunk(self, r: Request, budget: Optional[int]) -> int:
        """Prompt tokens of ``r`` to schedule now (0: none fit)."""
        n = len(r.prompt) - r.pos
        if self.prefill_chunk is not None:
            n = min(n, self.prefill_chunk)
            if budget is not None:
                n = min(n, budget)                                            # a chunk may shrink to what is left
        elif budget is not None and n > budget:
            n = 0                                                             # whole prompts only
        return max(n, 0)

    def head_admissible(self) -> bool:
        """Would the NEXT plan() admit the head of the waiting queue?  (A free slot is not enough: under ``max_step_tokens`` the
        prompt -- or, with ``prefill_chunk``, its first chunk -- must fit next to the running requests' decode tokens.)"""
        if not self.waiting or not self._free:
            return False
        budget = None
        if self.max_step_tokens is not None:
            budget = self.max_step_tokens - sum(1 for r in self.running if r.pos >= len(r.prompt))
            for r in self.running:                                            # prompts still entering go first, as in plan()
                if r.pos < len(r.prompt):
                    budget -= self._chunk(r, budget)
        return self._chunk(self.waiting[0], budget) > 0

    def plan(self) -> List[Item]:
        items = [Item(r, [r.out[-1]], r.pos) for r in self.running if r.pos >= len(r.prompt)]   # decode tokens first
        budget = None if self.max_step_tokens is None else self.max_step_tokens - len(items)
        for r in self.running:                                                # prompts still entering, admission order
            if r.pos < len(r.prompt):
                n = self._chunk(r, budget)
                if n > 0:
                    items.append(Item(r, r.prompt[r.pos:r.pos + n], r.pos))
                    if budget is not None:
                        budget -= n
        while self.waiting and self._free:
            r = self.waiting[0]
            n = self._chunk(r, budget)
            if n == 0:
                break                                                         # FIFO: no overtaking
            self.waiting.popleft()
            r.slot = self._free.pop()
            self.running.append(r)
            items.append(Item(r, r.prompt[:n], 0))
            if budget is not None:
                budget -= n
        return items

    def commit(self, items: List[Item], next_tokens: List[int]) -> List[Request]:
        """Record the token each scheduled request produced (a request whose prompt is still entering the
        cache produced none: the logits of a non-final chunk are dropped); returns the requests that finished."""
        done = []
        for it, t in zip(items, next_tokens):
            r = it.req
            r.pos = it.start + len(it.tokens)
            if r.pos < len(r.prompt):
                continue
            r.out.append(int(t))
            if len(r.out) == 1:
                r.t_first = time.perf_counter()
            if r.done:
                r.t_done = time.perf_counter()
                done.append(r)
        for r in done:
            self.running.remove(r)
            self._free.append(r.slot)
            self.finished[r.rid] = r
        return done


class ContinuousBatcher:
    def __init__(self, model: OneBitLlamaForCausalLM, max_batch: int = 32, max_len: int = 256,
                 max_step_tokens: Optional[int] = None, use_graph: bool = True, native: bool = True,
                 prefill_chunk: Optional[int] = None, max_burst: int = 1, long_context_from: int = 128, attn_chunk: int = 512):
        """``long_context_from``: a decode-only step whose longest context exceeds it replays the graph of the KEY-BLOCK attention
        (``onebit_decode_step_batched`` with ``attn_splits``: ``attn_chunk`` positions per workgroup, one graph per power-of-two
        split count) instead of the one-workgroup-per-(head, slot) form -- measured at 32 slots on 7B: equal at 128 cached
        tokens (2.37 ms), 2.64 vs 2.81 ms at 256, 3.23 vs 3.61 at 512 (tools/serve_ctx_probe.py), and the only form once 4 * max_len + 5.4 KB of scores no longer fit the LDS."""
        p = model.lm_head.weight
        if not p.is_cuda:
            raise RuntimeError("ContinuousBatcher needs the model on a ROCm GPU (no CPU fallback)")
        self.model, self.cfg, self.dev, self.dtype = model, model.config, p.device, p.dtype
        if max_len > model.config.max_position_embeddings:
            raise ValueError(f"max_len {max_len} exceeds max_position_embeddings {model.config.max_position_embeddings} "
                             "(the rope tables have that many rows)")
        self.sched = Scheduler(max_batch, max_len, max_step_tokens, prefill_chunk)
        cfg = self.cfg
        shape = (max_batch, cfg.num_key_value_heads, max_len, cfg.head_dim)
        self.cache = [(torch.zeros(shape, device=self.dev, dtype=self.dtype),
                       torch.zeros(shape, device=self.dev, dtype=self.dtype)) for _ in range(cfg.num_hidden_layers)]
        self.cos, self.sin = model._rope_tables(self.dev, self.dtype)
        self.steps = 0
        self.tokens_scheduled = 0
        # decode-only steps (the steady state) run as ONE HIP graph over static shapes: row i = slot i,
        # idle slots compute on token 0 at position 0 of their own (unused) cache slot
        self.use_graph = use_graph
        self._graph = None
        # per-step host -> device traffic of the graph step: ONE pinned staging row [ids | positions | native positions]
        # and one asynchronous copy (three pageable tensors + three blocking copies were ~0.15 ms of a 2 ms step)
        self._g_stage = torch.zeros(3 * max_batch, dtype=torch.long, device=self.dev)
        self._h_stage = torch.zeros(3 * max_batch, dtype=torch.long).pin_memory()
        self._h_stage_np = self._h_stage.numpy()
        self._g_ids = self._g_stage[:max_batch]
        self._g_pos = self._g_stage[max_batch:2 * max_batch]
        self._g_pos_native = self._g_stage[2 * max_batch:]              # -1 = idle slot; converted to int32 inside the graph
        self._g_next = torch.zeros(max_batch, dtype=torch.long, device=self.dev)
        self._h_next = torch.zeros(max_batch, dtype=torch.long).pin_memory()
        # decode bursts: while the set of running requests cannot change (nothing to admit, nobody finishes), up to
        # `max_burst` steps are enqueued back to back -- each step's tokens feed the next ON THE DEVICE -- and the host
        # reads all of them after ONE synchronisation.  Same tokens as step-by-step execution.  OFF by default (max_burst = 1):
        # measured on 7B at 32 slots the host is not what bounds the loop (2.19 ms per step with a sync per step, 2.17 /
        # 2.21 in bursts of 4 / 16, tools/burst_probe_serve.py) -- the step itself grows with the cached length (attention
        # over 1024 (head, slot) workgroups: 2.00 ms at position 16, 2.18 at 36-84).
        self.max_burst = max(1, int(max_burst))
        self._graph_fb = None
        self._g_ring = torch.zeros(self.max_burst, max_batch, dtype=torch.long, device=self.dev)
        self._h_ring = torch.zeros(self.max_burst, max_batch, dtype=torch.long).pin_memory()
        self.graph_steps = 0
        # native batched step (onebit_decode_step_batched) for the decode-only graph; torch-op glue otherwise
        self._native = None
        self._long = {}                         # split count -> (BatchedDecodeStep with attn_splits, its HIP graph)
        self.long_context_from, self.attn_chunk = int(long_context_from), int(attn_chunk)
        self._short_ok = 512 + 3 * 128 * 2 + 8 * 128 * 4 + 4 * max_len <= 64 * 1024       # the one-workgroup form's LDS bound
        self.long_steps = 0
        if native and use_graph and 2 <= max_batch <= 64 and self.dtype == torch.float16 and cfg.head_dim <= 128:
            from .engine import BatchedDecodeStep
            try:
                if self._short_ok:
                    self._native = BatchedDecodeStep(model, self.cache, max_batch, max_len)
                else:
                    self._native = self._long_engine(-(-max_len // self.attn_chunk))
            except ValueError:                  # shapes the C step does not take (e.g. in_features % 32 != 0)
                self._native = None
        # native MIXED step (onebit_mixed_step): every step that carries prompt tokens -- the workload that defines BASELINE
        # config 5 -- runs on the HIP kernels (one GEMM per projection over all scheduled rows, fused row glue, ragged attention,
        # lm_head on the sampling rows); `_forward` below (torch glue in the reference's op order) remains for native=False and
        # for checkpoints the native step refuses (head_dim other than 64 / 128, in_features % 32 != 0, fp32 parameters)
        self._mixed = None
        self.mixed_steps = 0
        self.time_mixed = self.time_decode = 0.0     # wall seconds in steps with / without prompt tokens (each step ends in a host sync)
        if native and self.dtype == torch.float16:
            from .engine import MixedStep
            try:
                self._mixed = MixedStep(model, self.cache, max_batch, max_len, max_rows=max_step_tokens or 1024)
                self._h_mixed = torch.zeros(max_batch, dtype=torch.int32).pin_memory()
            except ValueError:
                self._mixed = None

    def _long_engine(self, splits: int):
        """The native step with key-block attention for ``splits`` splits of ``attn_chunk`` positions (built on first use)."""
        from .engine import BatchedDecodeStep
        if splits not in self._long:
            self._long[splits] = [BatchedDecodeStep(self.model, self.cache, self.sched.max_batch, self.sched.max_len,
                                                    attn_splits=splits, attn_chunk=self.attn_chunk), None]
        return self._long[splits][0]

    def _engine_for(self, ctx: int):
        """(engine, graph key) for a decode-only step whose longest context is ``ctx`` tokens; key 0 = the short form."""
        if self._native is None:
            return None, 0
        if self._short_ok and (ctx <= self.long_context_from or self.long_context_from <= 0):
            return self._native, 0
        ns = 1
        while ns * self.attn_chunk < ctx:
            ns *= 2
        ns = min(ns, max(1, 1 << (-(-self.sched.max_len // self.attn_chunk) - 1).bit_length()))
        try:
            return self._long_engine(ns), ns
        except ValueError:
            return self._native, 0

    @torch.no_grad()
    def _decode_static(self, eng=None):
        """One token for every slot (static shapes, no host-side shape dependence): the same
        arithmetic as the batched decode branch of ``_forward`` with Lmax = max_len."""
        eng = eng if eng is not None else self._native
        if eng is not None:
            # idle slots keep pos = -1: their rows are computed, attention / cache append skipped
            eng.tokens.copy_(self._g_ids)
            eng.pos.copy_(self._g_pos_native)
            x = eng.launch()
            if eng.next_tokens is not None:                       # lm_head + argmax ran inside the C step
                self._g_next.copy_(eng.next_tokens)
            else:
                self._g_next.copy_((x @ self.model.lm_head.weight.t()).float().argmax(-1))
            return
        cfg, m = self.cfg, self.model.model
        H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        B, Lm = self.sched.max_batch, self.sched.max_len
        pos = self._g_pos
        h = m.embed_tokens(self._g_ids)
        c, s = self.cos[pos][:, None, :], self.sin[pos][:, None, :]
        rows = torch.arange(B, device=self.dev)
        hk = torch.arange(Hkv, device=self.dev)
        mask = (torch.arange(Lm, device=self.dev)[None, :] > pos[:, None])[:, None, None, :]
        for layer, (kc, vc) in zip(m.layers, self.cache):
            att = layer.self_attn
            x = layer.input_layernorm(h)
            q = att.q_proj(x).view(-1, H, D)
            k = att.k_proj(x).view(-1, Hkv, D)
            v = att.v_proj(x).view(-1, Hkv, D)
            q = (q * c) + (_rotate_half(q) * s)
            k = (k * c) + (_rotate_half(k) * s)
            kc[rows[:, None], hk[None, :], pos[:, None]] = k
            vc[rows[:, None], hk[None, :], pos[:, None]] = v
            keys, vals = kc, vc
            if Hkv != H:
                keys = keys.repeat_interleave(H // Hkv, dim=1)
                vals = vals.repeat_interleave(H // Hkv, dim=1)
            w = torch.matmul(q[:, :, None, :], keys.transpose(2, 3)) / math.sqrt(D)
            w = w.masked_fill(mask, float("-inf"))
            w = nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
            o = torch.matmul(w, vals).reshape(B, H * D)
            h = h + att.o_proj(o)
            h = h + layer.mlp(layer.post_attention_layernorm(h))
        self._g_next.copy_(self.model.lm_head(m.norm(h)).float().argmax(-1))

    def _graph_step(self, items: List[Item], sync: bool = True) -> Optional[List[int]]:
        B = self.sched.max_batch
        st = self._h_stage_np
        st[:2 * B] = 0
        st[2 * B:] = -1
        for it in items:
            sl = it.req.slot
            st[sl], st[B + sl], st[2 * B + sl] = it.tokens[0], it.start, it.start
        self._g_stage.copy_(self._h_stage, non_blocking=True)            # stream-ordered before the replay
        eng, key = self._engine_for(max(it.start for it in items) + 1)
        graph = self._graph if key == 0 else self._long[key][1]
        if graph is None:
            self._decode_static(eng)                                # warm-up (allocations, lazy init): the step itself, idempotent
            torch.cuda.synchronize(self.dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._decode_static(eng)
            if key == 0:
                self._graph = graph
            else:
                self._long[key][1] = graph
        graph.replay()
        self.graph_steps += 1
        self.long_steps += key != 0
        if not sync:
            return None
        self._h_next.copy_(self._g_next, non_blocking=True)
        torch.cuda.current_stream(self.dev).synchronize()                # the one host sync of the step (also fences the staging row)
        nxt = self._h_next.tolist()
        return [nxt[it.req.slot] for it in items]

    def _burst_len(self, items: List[Item]) -> int:
        """Steps that may run without the host in the loop: decode-only, native step, nothing admissible, and no request
        of the step finishes (or reaches the end of its cache slot) before the last of them."""
        if self._native is None or self._native.next_tokens is None or not self.use_graph or self.max_burst <= 1:
            return 1
        if self.sched.head_admissible():         # (a waiting request that max_step_tokens keeps out does not end the burst)
            return 1
        if len(items) != len(self.sched.running):
            return 1
        if self._engine_for(max(it.start for it in items) + self.max_burst + 1)[1] != 0:
            return 1                             # (bursts feed tokens back through the short-form engine's buffers only)
        n = min(it.req.max_new_tokens - len(it.req.out) for it in items)
        n = min(n, min(self.sched.max_len - it.start for it in items), self.max_burst)
        return max(1, n)

    @torch.no_grad()
    def _feedback_static(self):
        """The step after a graph step of the SAME requests: token <- the token just produced, position + 1, on the device."""
        nat = self._native
        nat.tokens.copy_(nat.next_tokens)
        nat.pos.add_((nat.pos >= 0).to(nat.pos.dtype))
        nat.launch()
        self._g_next.copy_(nat.next_tokens)

    def _graph_burst(self, items: List[Item], n: int) -> List[List[int]]:
        """n consecutive decode steps of `items`' requests; returns the n token lists (in `items` order)."""
        first = self._graph_step(items, sync=False)          # staged from the host; leaves its tokens in _g_next
        assert first is None
        self._g_ring[0].copy_(self._g_next, non_blocking=True)
        if self._graph_fb is None:
            # the warm-up below runs one feedback step on the LIVE state and rewinds it: it writes KV row start + 1 of every
            # slot, which the replay rewrites identically -- provided that row exists and the burst does replay it
            assert n >= 2 and max(it.start for it in items) + 1 < self.sched.max_len, "burst warm-up outside the cache slot"
            tok0, pos0, nxt0 = self._native.tokens.clone(), self._native.pos.clone(), self._native.next_tokens.clone()
            torch.cuda.synchronize(self.dev)
            self._feedback_static()                                  # warm-up on the live state ...
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._feedback_static()
            self._graph_fb = g
            # ... then rewind it (the warm-up wrote one more KV row per slot: rewritten identically by the replay)
            self._native.tokens.copy_(tok0); self._native.pos.copy_(pos0); self._native.next_tokens.copy_(nxt0)
        for j in range(1, n):
            self._graph_fb.replay()
            self._g_ring[j].copy_(self._g_next, non_blocking=True)
        self.graph_steps += n - 1
        self._h_ring[:n].copy_(self._g_ring[:n], non_blocking=True)
        torch.cuda.current_stream(self.dev).synchronize()
        rows = self._h_ring[:n].tolist()
        return [[row[it.req.slot] for it in items] for row in rows]

    def add_request(self, prompt: List[int], max_new_tokens: int) -> int:
        return self.sched.add(prompt, max_new_tokens)

    def _mixed_step(self, items: List[Item]) -> List[int]:
        """One step through ``onebit_mixed_step``: the greedy token after the last scheduled row of every item."""
        nxt = self._mixed.launch([(it.req.slot, it.start, it.tokens) for it in items])
        self._h_mixed[:len(items)].copy_(nxt, non_blocking=True)
        torch.cuda.current_stream(self.dev).synchronize()                # the one host sync of the step
        self.mixed_steps += 1
        return self._h_mixed[:len(items)].tolist()

    @torch.no_grad()
    def _forward(self, items: List[Item]) -> torch.Tensor:
        """fp32 logits [len(items), vocab] of the last scheduled token of every item."""
        cfg, m = self.cfg, self.model.model
        H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        ids = torch.tensor([t for it in items for t in it.tokens], device=self.dev, dtype=torch.long)
        pos = torch.tensor([it.start + j for it in items for j in range(len(it.tokens))], device=self.dev)
        bounds, a = [], 0
        for it in items:
            bounds.append((a, a + len(it.tokens)))
            a += len(it.tokens)
        dec = [i for i, it in enumerate(items) if len(it.tokens) == 1]
        dec_rows = None
        if len(dec) > 1:
            dec_rows = torch.tensor([bounds[i][0] for i in dec], device=self.dev)
            dec_slots = torch.tensor([items[i].req.slot for i in dec], device=self.dev)
            dec_start = torch.tensor([items[i].start for i in dec], device=self.dev)
            Lmax = max(items[i].start for i in dec) + 1
            dec_mask = (torch.arange(Lmax, device=self.dev)[None, :] > dec_start[:, None])[:, None, None, :]
            hk = torch.arange(Hkv, device=self.dev)
        h = m.embed_tokens(ids)                                               # [T, hidden]
        c, s = self.cos[pos][:, None, :], self.sin[pos][:, None, :]           # [T, 1, D]
        for layer, (kc, vc) in zip(m.layers, self.cache):
            att = layer.self_attn
            x = layer.input_layernorm(h)
            q = att.q_proj(x).view(-1, H, D)                                  # ONE call per projection for all tokens
            k = att.k_proj(x).view(-1, Hkv, D)
            v = att.v_proj(x).view(-1, Hkv, D)
            q = (q * c) + (_rotate_half(q) * s)
            k = (k * c) + (_rotate_half(k) * s)
            o = torch.empty(ids.shape[0], H * D, device=self.dev, dtype=self.dtype)
            if dec_rows is not None:
                # all single-token items at once: padded to the longest context, positions beyond a
                # request's own length masked to -inf (probability exactly 0, so the result equals
                # the unpadded computation of LlamaAttentionInf.forward)
                kc[dec_slots[:, None], hk[None, :], dec_start[:, None]] = k[dec_rows]
                vc[dec_slots[:, None], hk[None, :], dec_start[:, None]] = v[dec_rows]
                keys, vals = kc[dec_slots, :, :Lmax], vc[dec_slots, :, :Lmax]          # [Bd, Hkv, Lmax, D]
                if Hkv != H:
                    keys = keys.repeat_interleave(H // Hkv, dim=1)
                    vals = vals.repeat_interleave(H // Hkv, dim=1)
                w = torch.matmul(q[dec_rows][:, :, None, :], keys.transpose(2, 3)) / math.sqrt(D)   # [Bd, H, 1, Lmax]
                w = w.masked_fill(dec_mask, float("-inf"))
                w = nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
                o[dec_rows] = torch.matmul(w, vals).reshape(-1, H * D)
            for it, (a, b) in zip(items, bounds):
                if b - a == 1 and dec_rows is not None:
                    continue
                sl, n, L = it.req.slot, b - a, it.start + b - a
                kc[sl, :, it.start:L] = k[a:b].transpose(0, 1)
                vc[sl, :, it.start:L] = v[a:b].transpose(0, 1)
                keys, vals = kc[sl, :, :L], vc[sl, :, :L]                     # [Hkv, L, D]
                if Hkv != H:
                    keys = keys.repeat_interleave(H // Hkv, dim=0)
                    vals = vals.repeat_interleave(H // Hkv, dim=0)
                w = torch.matmul(q[a:b].transpose(0, 1), keys.transpose(1, 2)) / math.sqrt(D)     # [H, n, L]
                if n > 1:
                    mask = torch.full((n, L), torch.finfo(w.dtype).min, device=self.dev, dtype=w.dtype)
                    w = w + torch.triu(mask, diagonal=it.start + 1)[None]
                w = nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
                o[a:b] = torch.matmul(w, vals).transpose(0, 1).reshape(n, H * D)
            h = h + att.o_proj(o)
            h = h + layer.mlp(layer.post_attention_layernorm(h))
        last = torch.tensor([b - 1 for _, b in bounds], device=self.dev)
        return self.model.lm_head(m.norm(h[last])).float()

    def step(self) -> List[Request]:
        """Schedule and run one step; returns the requests that finished in it."""
        items = self.sched.plan()
        if not items:
            return []
        self.steps += 1
        self.tokens_scheduled += sum(len(it.tokens) for it in items)
        t0 = time.perf_counter()
        try:
            return self._run_items(items)
        finally:
            if all(len(it.tokens) == 1 for it in items):
                self.time_decode += time.perf_counter() - t0
            else:
                self.time_mixed += time.perf_counter() - t0

    def _run_items(self, items: List[Item]) -> List[Request]:
        if self.use_graph and all(len(it.tokens) == 1 for it in items):
            n = self._burst_len(items)
            if n <= 1:
                return self.sched.commit(items, self._graph_step(items))
            rows = self._graph_burst(items, n)
            self.steps += n - 1
            self.tokens_scheduled += (n - 1) * len(items)
            done = self.sched.commit(items, rows[0])
            for j in range(1, n):                    # the plans the scheduler would have made, committed in order
                nxt = [Item(it.req, [rows[j - 1][i]], it.start + j) for i, it in enumerate(items)]
                done = done + self.sched.commit(nxt, rows[j])
            return done
        if self._mixed is not None:
            return self.sched.commit(items, self._mixed_step(items))
        logits = self._forward(items)
        return self.sched.commit(items, logits.argmax(-1).tolist())

    def run(self) -> Dict[int, List[int]]:
        """Drain the queue; {request id: generated tokens}."""
        while not self.sched.idle:
            before = (len(self.sched.waiting), sum(r.pos + len(r.out) for r in self.sched.running), len(self.sched.finished))
            self.step()
            after = (len(self.sched.waiting), sum(r.pos + len(r.out) for r in self.sched.running), len(self.sched.finished))
            if after == before:            # an empty plan with work pending would spin forever
                raise RuntimeError("ContinuousBatcher.run: no progress (a waiting prompt exceeds the step budget?)")
        return {rid: r.out for rid, r in self.sched.finished.items()}
