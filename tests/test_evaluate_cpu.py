"""Host logic of the evaluation callers (onebit_amd/evaluate.py) on a CPU stand-in model: request
grouping / ordering / padding / left truncation of loglikelihood_tokens, window and limit
arithmetic of perplexity -- each against a direct, unbatched computation."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from onebit_amd.evaluate import loglikelihood_tokens, perplexity


class CausalStub(torch.nn.Module):
    """logits[t] depend on tokens 0..t only (running mean of embeddings), so right padding is harmless."""

    def __init__(self, vocab=50, dim=16, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.emb = torch.nn.Parameter(torch.randn(vocab, dim, generator=g), requires_grad=False)
        self.out = torch.nn.Parameter(torch.randn(dim, vocab, generator=g), requires_grad=False)
        self.calls = []

    def forward(self, ids):
        self.calls.append(tuple(ids.shape))
        h = self.emb[ids].cumsum(1) / torch.arange(1, ids.shape[1] + 1).view(1, -1, 1)
        return h @ self.out


def _direct(model, ctx, cont, max_length):
    inp = torch.tensor((list(ctx) + list(cont))[-(max_length + 1):][:-1]).unsqueeze(0)
    lp = F.log_softmax(model(inp), dim=-1)[0]
    lg = lp[inp.shape[1] - len(cont):]
    c = torch.tensor(cont)
    return float(lg.gather(1, c.unsqueeze(-1)).sum()), bool((lg.argmax(-1) == c).all())


def test_loglikelihood_matches_unbatched_and_orders_chunks():
    m = CausalStub()
    g = torch.Generator().manual_seed(1)
    reqs = [(torch.randint(1, 50, (a,), generator=g).tolist(), torch.randint(1, 50, (b,), generator=g).tolist())
            for a, b in [(3, 2), (10, 4), (1, 1), (25, 12), (6, 6), (2, 9), (8, 1)]]
    expect = [_direct(m, c, t, 16) for c, t in reqs]
    m.calls.clear()
    got = loglikelihood_tokens(m, reqs, batch_size=3, max_length=16)
    assert len(got) == len(reqs)
    for (a, fa), (b, fb) in zip(got, expect):
        assert abs(a - b) < 1e-4 and fa == fb
    # chunks are formed in descending length order and padded to their first (longest) member;
    # the 37-token request is left-truncated to max_length
    assert m.calls == [(3, 16), (3, 10), (1, 1)]


def test_loglikelihood_groups_identical_token_sequences():
    m = CausalStub(seed=3)
    a = ([5, 6, 7], [8, 9])
    b = ([5, 6, 7, 8], [9])              # same tokens, different split: shares a's answer (Reorderer quirk)
    got = loglikelihood_tokens(m, [a, b, a], batch_size=8, max_length=16)
    ref = _direct(m, *a, 16)
    assert all(abs(x[0] - ref[0]) < 1e-5 and x[1] == ref[1] for x in got)
    assert m.calls[-1] == (1, 4)         # ONE row was evaluated for the three requests


def test_loglikelihood_greedy_flag_true_for_argmax_continuation():
    m = CausalStub(seed=5)
    ctx = [4, 9, 2]
    cont = []
    for _ in range(3):
        cont.append(int(m(torch.tensor([ctx + cont]))[0, -1].argmax()))
    (ll, greedy), = loglikelihood_tokens(m, [(ctx, cont)], batch_size=1, max_length=16)
    assert greedy and ll < 0


def test_perplexity_windows_limit_and_scaling():
    m = CausalStub(seed=7)
    toks = torch.randint(0, 50, (1, 75), generator=torch.Generator().manual_seed(2))
    S = 16                                # 4 windows, 11 tokens dropped
    nll = []
    for i in range(4):
        w = toks[:, i * S:(i + 1) * S]
        lp = F.log_softmax(m(w)[0, :-1], dim=-1)
        nll.append(float(-lp.gather(1, w[0, 1:].unsqueeze(-1)).mean()) * S)     # mean over S-1, times S
    assert abs(perplexity(m, toks, S) - math.exp(sum(nll) / (4 * S))) < 1e-3
    # limit = 1: windows 0 and 1 are evaluated, the divisor stays nsamples * seqlen (lm_eval.py:123-126)
    assert abs(perplexity(m, toks, S, limit=1) - math.exp(sum(nll[:2]) / (4 * S))) < 1e-3
    with np.testing.assert_raises(ValueError):
        perplexity(m, toks[:, :10], S)


# ---- data-parallel evaluation over a process group (gloo, world 2 and 3): every rank returns what one process returns
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _eval_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = CausalStub(seed=3)
        g = torch.Generator().manual_seed(11)
        ids = torch.randint(0, 50, (1, 7 * 12 + 5), generator=g)
        reqs = []
        for n, m in [(5, 3), (9, 1), (2, 6), (14, 2), (5, 3), (3, 3), (8, 4), (1, 1), (6, 5), (11, 2), (4, 4)]:
            reqs.append((torch.randint(0, 50, (n,), generator=g).tolist(), torch.randint(0, 50, (m,), generator=g).tolist()))
        reqs[4] = reqs[0]                                       # a duplicate group
        ref_ppl = [perplexity(model, ids, 12), perplexity(model, ids, 12, limit=3)]
        ref_ll = loglikelihood_tokens(model, reqs, batch_size=3, max_length=10)
        model.calls.clear()
        got_ppl = [perplexity(model, ids, 12, rank=rank, world=world), perplexity(model, ids, 12, limit=3, rank=rank, world=world)]
        ncalls_ppl = len(model.calls)
        model.calls.clear()
        got_ll = loglikelihood_tokens(model, reqs, batch_size=3, max_length=10, rank=rank, world=world)
        res = [None] * world
        dist.all_gather_object(res, (got_ppl == ref_ppl, got_ll == ref_ll, ncalls_ppl, len(model.calls)))
        if rank == 0:
            out.put(res)
    finally:
        dist.destroy_process_group()


def test_data_parallel_evaluation_equals_single_process():
    """perplexity windows / loglikelihood chunks dealt round-robin to the ranks of a gloo group: every rank gets the
    single-process results bit for bit, and the model calls are actually split between the ranks."""
    import pytest
    import torch.multiprocessing as mp
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = q.get(timeout=180)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(r[0] and r[1] for r in res), res
        # 7 windows + 4 windows (limit = 3) and 4 chunks (10 groups of 3) split over the ranks
        assert sum(r[2] for r in res) == 7 + 4 and max(r[2] for r in res) < 11
        assert sum(r[3] for r in res) == 4 and max(r[3] for r in res) <= 2
