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
