"""Scheduler of the continuous batcher (pure Python): admission, plans, completion, slot reuse."""
import pytest

from onebit_amd.serving import Scheduler


def test_scheduler_mixed_prefill_decode_plans_and_slot_reuse():
    s = Scheduler(max_batch=2, max_len=16)
    a = s.add([1, 2, 3], 2)
    b = s.add([4], 3)
    c = s.add([5, 6], 1)
    p1 = s.plan()                                   # a and b admitted (prefill), c waits
    assert [(it.req.rid, it.tokens, it.start) for it in p1] == [(a, [1, 2, 3], 0), (b, [4], 0)]
    assert s.commit(p1, [10, 20]) == []
    p2 = s.plan()                                   # both decode; no free slot for c
    assert [(it.req.rid, it.tokens, it.start) for it in p2] == [(a, [10], 3), (b, [20], 1)]
    done = s.commit(p2, [11, 21])
    assert [r.rid for r in done] == [a] and s.finished[a].out == [10, 11]
    p3 = s.plan()                                   # b decodes, c is admitted into a's slot in the same step
    assert [(it.req.rid, it.tokens, it.start) for it in p3] == [(b, [21], 2), (c, [5, 6], 0)]
    assert p3[1].req.slot == s.finished[a].slot
    done = s.commit(p3, [22, 50])
    assert sorted(r.rid for r in done) == [b, c] and s.idle
    assert s.finished[b].out == [20, 21, 22] and s.finished[c].out == [50]
    assert s.plan() == []


def test_scheduler_token_budget_is_fifo_and_validates():
    s = Scheduler(max_batch=4, max_len=32, max_step_tokens=6)
    s.add([1] * 4, 2)
    s.add([2] * 4, 2)                               # would exceed the 6-token budget in step 1
    s.add([3], 2)                                   # must not overtake request 1
    p = s.plan()
    assert [it.req.rid for it in p] == [0]
    s.commit(p, [9])
    p = s.plan()                                    # 1 decode token + 4 prompt tokens + 1 prompt token
    assert [(it.req.rid, len(it.tokens)) for it in p] == [(0, 1), (1, 4), (2, 1)]
    with pytest.raises(ValueError):
        s.add([], 1)
    with pytest.raises(ValueError):
        s.add([1] * 40, 1)
    with pytest.raises(ValueError):
        s.add([1] * 7, 1)


def test_scheduler_chunked_prefill_plans_and_first_token():
    """prefill_chunk: a prompt enters at most that many tokens per step, running requests keep decoding in the
    same steps, and a request produces its first token only with its last chunk."""
    s = Scheduler(max_batch=2, max_len=32, prefill_chunk=3)
    a = s.add([1, 2, 3, 4, 5, 6, 7], 2)             # 3 chunks: 3 + 3 + 1
    b = s.add([9], 3)
    p1 = s.plan()
    assert [(it.req.rid, it.tokens, it.start) for it in p1] == [(a, [1, 2, 3], 0), (b, [9], 0)]
    assert s.commit(p1, [100, 20]) == []            # a's logits are dropped: its prompt is still entering
    assert s.running[0].out == [] and s.running[0].pos == 3
    p2 = s.plan()                                   # b decodes first, a's next chunk behind it
    assert [(it.req.rid, it.tokens, it.start) for it in p2] == [(b, [20], 1), (a, [4, 5, 6], 3)]
    s.commit(p2, [21, 101])
    p3 = s.plan()
    assert [(it.req.rid, it.tokens, it.start) for it in p3] == [(b, [21], 2), (a, [7], 6)]
    done = s.commit(p3, [22, 30])                   # a's last chunk: first token sampled; b finishes
    assert [r.rid for r in done] == [b] and s.finished[b].out == [20, 21, 22]
    p4 = s.plan()
    assert [(it.req.rid, it.tokens, it.start) for it in p4] == [(a, [30], 7)]
    done = s.commit(p4, [31])
    assert [r.rid for r in done] == [a] and s.finished[a].out == [30, 31] and s.idle


def test_scheduler_chunked_prefill_with_token_budget():
    """With a step budget the chunk shrinks to what is left, long prompts are accepted, FIFO order holds."""
    s = Scheduler(max_batch=3, max_len=64, max_step_tokens=5, prefill_chunk=4)
    a = s.add(list(range(10)), 1)                   # longer than the budget: fine when chunked
    b = s.add([7, 8, 9], 1)
    p = s.plan()                                    # 4 tokens of a, then 1 of b (budget 5)
    assert [(it.req.rid, len(it.tokens), it.start) for it in p] == [(a, 4, 0), (b, 1, 0)]
    s.commit(p, [0, 0])
    p = s.plan()
    assert [(it.req.rid, len(it.tokens), it.start) for it in p] == [(a, 4, 4), (b, 1, 1)]
    s.commit(p, [0, 0])
    p = s.plan()                                    # a: last 2 tokens; b: its last token
    assert [(it.req.rid, len(it.tokens), it.start) for it in p] == [(a, 2, 8), (b, 1, 2)]
    done = s.commit(p, [41, 42])
    assert sorted(r.rid for r in done) == [a, b] and s.finished[a].out == [41] and s.finished[b].out == [42]
    with pytest.raises(ValueError):
        Scheduler(2, 8, prefill_chunk=0)


def test_scheduler_rejects_budgets_that_cannot_progress():
    """A step budget that leaves no room for progress is refused at construction (round-2 review): zero / negative
    budgets, and with chunked prefill a budget that every decoding slot's token can exhaust."""
    import pytest
    with pytest.raises(ValueError):
        Scheduler(max_batch=4, max_len=32, max_step_tokens=0)
    with pytest.raises(ValueError):
        Scheduler(max_batch=4, max_len=32, max_step_tokens=4, prefill_chunk=2)
    s = Scheduler(max_batch=4, max_len=32, max_step_tokens=5, prefill_chunk=2)
    for _ in range(5):
        s.add([1, 2, 3, 4, 5, 6, 7], 3)
    steps = 0
    while not s.idle:                               # every step makes progress: no starvation of entering prompts
        items = s.plan()
        assert items and sum(len(i.tokens) for i in items) <= 5
        s.commit(items, [9] * len(items))
        steps += 1
        assert steps < 200
    assert len(s.finished) == 5


def test_head_admissible_agrees_with_plan():
    """`head_admissible` (what ends a host-free decode burst, serving.py `_burst_len`) is exactly "the next plan() admits the head
    of the queue": a free slot alone does not make a request admissible under max_step_tokens (advisor finding, round 4)."""
    s = Scheduler(max_batch=4, max_len=64, max_step_tokens=8)
    a = s.add(list(range(6)), 4)
    assert s.head_admissible()
    items = s.plan()
    s.commit(items, [1] * len(items))
    b = s.add(list(range(8)), 2)                  # 8 prompt tokens + 1 running decode token > 8: blocked although 3 slots are free
    assert s._free and s.waiting and not s.head_admissible()
    items = s.plan()
    assert [it.req.rid for it in items] == [a]
    s.commit(items, [1])
    c = Scheduler(max_batch=2, max_len=64, max_step_tokens=4, prefill_chunk=3)
    c.add(list(range(7)), 2)
    c.add(list(range(5)), 2)
    for _ in range(6):                            # every step: the prediction equals what plan() then does
        n_wait = len(c.waiting)
        pred = c.head_admissible()
        items = c.plan()
        assert pred == (len(c.waiting) < n_wait)
        c.commit(items, [1] * len(items))
    assert not Scheduler(max_batch=1, max_len=8).head_admissible()       # nothing waiting
