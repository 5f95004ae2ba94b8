"""Shared-memory ring and SharedMemoryQueue (host side, CPU only): FIFO order, blocking
semantics, size limits, multi-process producers/consumers."""

import multiprocessing as mp
import queue
import time

import pytest


def test_ring_roundtrip_and_limits(libprl):
    from pipelinerl_amd.ring import Ring

    r = Ring(n_slots=4, slot_bytes=64)
    assert (r.n_slots, r.slot_bytes) == (4, 64)
    for i in range(4):
        r.put_bytes(bytes([i]) * (i + 1))
    assert r.qsize() == 4
    with pytest.raises(queue.Full):
        r.put_bytes(b"x", block=False)
    t0 = time.time()
    with pytest.raises(queue.Full):
        r.put_bytes(b"x", timeout=0.05)
    assert time.time() - t0 >= 0.04
    assert [r.get_bytes() for _ in range(4)] == [bytes([i]) * (i + 1) for i in range(4)]
    with pytest.raises(queue.Empty):
        r.get_bytes(block=False)
    with pytest.raises(ValueError):
        r.put_bytes(b"y" * 65)
    r.put_bytes(b"")  # empty records are legal
    assert r.get_bytes() == b""
    assert r.max_record_bytes() == 4
    r.close()


def _producer(q, start, n):
    for i in range(start, start + n):
        q.put({"i": i, "payload": list(range(i % 7))})


def _consumer(q, out, n):
    for _ in range(n):
        out.put(q.get(timeout=20)["i"])


@pytest.mark.parametrize("method", ["fork", "spawn"])
def test_shared_memory_queue_multiprocess(libprl, method):
    """The reference's use: N worker processes consume chunks and produce results
    (preprocess.py:489-490).  Works for fork (inherited handle) and spawn (re-attach by name)."""
    from pipelinerl_amd.shared_memory_array import SharedMemoryQueue

    ctx = mp.get_context(method)
    q = SharedMemoryQueue(None, max_size=8, max_entry_size=4096)
    out = ctx.Queue()
    n_prod, n_cons, per = 3, 2, 40
    procs = [ctx.Process(target=_producer, args=(q, k * per, per)) for k in range(n_prod)]
    cons = [ctx.Process(target=_consumer, args=(q, out, n_prod * per // n_cons)) for _ in range(n_cons)]
    for p in procs + cons:
        p.start()
    got = sorted(out.get(timeout=30) for _ in range(n_prod * per))
    for p in procs + cons:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got == list(range(n_prod * per))
    assert q.qsize() == 0 and not q.full()
    assert 0 < q.max_actual_entry_size() <= 4096
    q.close()


def test_shared_memory_queue_interface(libprl):
    from pipelinerl_amd.shared_memory_array import SharedMemoryQueue

    q = SharedMemoryQueue(None, max_size=2, max_entry_size=128)
    q.put([1, 2, 3])
    q.put("x")
    assert q.full() and q.qsize() == 2
    with pytest.raises(queue.Full):
        q.put(1, block=False)
    with pytest.raises(ValueError):
        q.put(b"z" * 1000)
    assert q.get() == [1, 2, 3] and q.get() == "x"
    with pytest.raises(queue.Empty):
        q.get(timeout=0.01)
    assert q.get_memory_size() >= 256
    with pytest.raises(ValueError):
        SharedMemoryQueue(None, 0, 10)
    q.close()
