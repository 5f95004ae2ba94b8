"""Golden operation trace of the reference's SharedMemoryQueue (pipelinerl/shared_memory_array.py:109-196),
produced by running the class itself (it imports cleanly here).

A scripted sequence of non-blocking / timed puts and gets on a 3-slot queue; after every operation
the script records what the caller can observe: the returned item or the exception class, `qsize()`,
`full()` and `max_actual_entry_size()`.  The reference's bookkeeping sits on two multiprocessing
queues whose feeder threads are asynchronous, so the script sleeps 50 ms after every operation
before observing (the values recorded are the settled ones).

    python tests/golden/make_queue_golden.py
"""

from __future__ import annotations

import json
import pickle
import sys
import time
from multiprocessing.managers import SharedMemoryManager
from pathlib import Path

HERE = Path(__file__).resolve().parent

# (op, argument)   put: item spec;  get: None.   Items are small picklable values.
SCRIPT = [
    ("get", None),                       # empty queue, non-blocking
    ("put", {"a": 1}),
    ("put", [1, 2, 3]),
    ("get", None),
    ("put", "x" * 40),
    ("put", {"nested": {"k": [1.5, None, True]}}),
    ("put", 7),                          # third slot filled -> full
    ("put", 8),                          # Full
    ("get", None),
    ("get", None),
    ("put", "y" * 400),                  # exceeds max_entry_size -> ValueError (and what happens to the slot)
    ("put", 9),
    ("put", 10),
    ("put", 11),
    ("get", None),
    ("get", None),
    ("get", None),
    ("get", None),
    ("get", None),
]


def run(queue_cls, smm, full_exc, empty_exc) -> list[dict]:
    q = queue_cls(smm, 3, 256)
    trace = []
    for op, arg in SCRIPT:
        rec: dict = {"op": op}
        try:
            if op == "put":
                rec["pickled_size"] = len(pickle.dumps(arg))
                q.put(arg, block=False)
                rec["result"] = "ok"
            else:
                rec["result"] = {"item": q.get(block=True, timeout=0.2)}
        except full_exc:
            rec["result"] = "Full"
        except empty_exc:
            rec["result"] = "Empty"
        except ValueError:
            rec["result"] = "ValueError"
        time.sleep(0.05)
        rec.update(qsize=q.qsize(), full=bool(q.full()), max_actual_entry_size=q.max_actual_entry_size())
        trace.append(rec)
    return trace


def main() -> None:
    sys.path.insert(0, "/root/reference")
    from queue import Empty, Full

    from pipelinerl.shared_memory_array import SharedMemoryQueue

    with SharedMemoryManager() as smm:
        trace = run(SharedMemoryQueue, smm, Full, Empty)
        ctor = {}
        for name, args in (("zero_size", (0, 16)), ("zero_entry", (4, 0))):
            try:
                SharedMemoryQueue(smm, *args)
                ctor[name] = "ok"
            except ValueError:
                ctor[name] = "ValueError"
        mem = SharedMemoryQueue(smm, 3, 256).get_memory_size()
    (HERE / "queue_trace.json").write_text(json.dumps({"trace": trace, "ctor": ctor, "memory_size_3x256": mem}, indent=1))
    for r in trace:
        print(r)
    print(ctor, mem)


if __name__ == "__main__":
    main()
