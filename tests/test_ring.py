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


def test_shared_memory_queue_matches_reference_trace(libprl):
    """Differential test against the operation trace of the reference's own SharedMemoryQueue
    (tests/golden/make_queue_golden.py): results, exceptions, qsize/full/max_actual_entry_size after
    every operation.  One deliberate divergence: the reference leaks a slot on an oversize put."""
    import json
    import pickle
    import sys
    from queue import Empty, Full

    from helpers import GOLDEN
    from pipelinerl_amd.shared_memory_array import SharedMemoryQueue

    sys.path.insert(0, str(GOLDEN))
    from make_queue_golden import SCRIPT

    g = json.loads((GOLDEN / "queue_trace.json").read_text())
    for name, args in (("zero_size", (0, 16)), ("zero_entry", (4, 0))):
        assert g["ctor"][name] == "ValueError"
        with pytest.raises(ValueError):
            SharedMemoryQueue(None, *args)
    q = SharedMemoryQueue(None, 3, 256)
    mine = []
    for op, arg in SCRIPT:
        try:
            if op == "put":
                q.put(arg, block=False)
                res = "ok"
            else:
                res = {"item": q.get(block=True, timeout=0.05)}
        except Full:
            res = "Full"
        except Empty:
            res = "Empty"
        except ValueError:
            res = "ValueError"
        mine.append({"op": op, "result": res, "qsize": q.qsize(), "full": q.full(), "max_actual_entry_size": q.max_actual_entry_size()})
    q.close()
    leak = next(i for i, r in enumerate(g["trace"]) if r["result"] == "ValueError")
    for i in range(leak + 1):  # identical up to and including the oversize put
        want = {k: g["trace"][i][k] for k in ("op", "result", "qsize", "full", "max_actual_entry_size")}
        assert mine[i] == want, (i, mine[i], want)
    assert len(pickle.dumps(SCRIPT[leak][1])) == g["trace"][leak]["pickled_size"] > 256
    # afterwards the reference runs with 2 usable slots (put/put -> ok/Full); this queue keeps all 3
    assert [r["result"] for r in g["trace"][leak + 1 : leak + 4]] == ["ok", "Full", "Full"]
    assert [r["result"] for r in mine[leak + 1 : leak + 4]] == ["ok", "ok", "Full"]
    assert [r["result"] for r in mine[leak + 4 :]] == [{"item": {"nested": {"k": [1.5, None, True]}}}, {"item": 9}, {"item": 10}, "Empty", "Empty"]
    assert mine[-1]["qsize"] == 0 and mine[-1]["full"] is False


def test_shared_memory_array_follows_the_reference_trace():
    """`SharedMemoryArray` (shared_memory_array.py:9-106) replayed against a trace of the reference class (tests/golden/make_array_golden.py):
    items, `None` for an empty slot, the oversize / index errors with the reference's messages, `max_actual_entry_size` after every call."""
    import json
    import multiprocessing as mp
    from multiprocessing.managers import SharedMemoryManager

    from helpers import GOLDEN
    from pipelinerl_amd.shared_memory_array import SharedMemoryArray

    g = json.loads((GOLDEN / "array_trace.json").read_text())
    for make in ("managed", "own_segment"):
        with SharedMemoryManager() as smm:
            arr = SharedMemoryArray(smm if make == "managed" else None, 4, 256)
            assert len(arr) == g["len"] and arr.get_memory_size() >= g["memory_size_at_least"]
            for (op, index, value), want in zip(g["script"], g["trace"]):
                if op == "set" and isinstance(value, str) and want.get("pickled_size") == 255:
                    value = bytes.fromhex(value)
                try:
                    if op == "set":
                        arr[index] = value
                        got = "ok"
                    else:
                        item = arr[index]
                        got = {"item": item.hex() if isinstance(item, bytes) else item, "bytes": isinstance(item, bytes)}
                except Exception as e:  # noqa: BLE001
                    got = {"raises": type(e).__name__, "message": str(e)}
                assert got == want["result"], (op, index)
                assert arr.max_actual_entry_size() == want["max_actual_entry_size"], (op, index)
            if make == "managed":  # a child process reads what the parent wrote (the object travels like the reference's)
                arr[2] = {"from": "parent"}
                ctx = mp.get_context("fork")
                q = ctx.Queue()
                p = ctx.Process(target=lambda a, out: out.put(a[2]), args=(arr, q))
                p.start()
                assert q.get(timeout=10) == {"from": "parent"}
                p.join(5)
            arr.close()
    for name, args in (("zero_entries", (0, 16)), ("zero_size", (4, 0))):
        with pytest.raises(ValueError, match=g["ctor_errors"][name]["message"]):
            SharedMemoryArray(None, *args)
