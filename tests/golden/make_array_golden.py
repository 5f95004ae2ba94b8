"""Golden operation trace of the reference's SharedMemoryArray (pipelinerl/shared_memory_array.py:9-106), produced by running
the class itself: a scripted sequence of item assignments and reads on a 4-slot array, recording what the caller observes
(returned item or exception class + message, `len()`, `get_memory_size()`, `max_actual_entry_size()`).

    python tests/golden/make_array_golden.py
"""

from __future__ import annotations

import json
import pickle
import sys
from multiprocessing.managers import SharedMemoryManager
from pathlib import Path

HERE = Path(__file__).resolve().parent

SCRIPT = [
    ("get", 0, None), ("set", 0, {"a": 1}), ("get", 0, None), ("set", 3, [1, 2, 3]), ("get", 3, None), ("get", 1, None),
    ("set", 1, "x" * 100), ("set", 1, "y" * 500), ("get", 1, None), ("set", 4, 1), ("get", -1, None), ("get", 7, None),
    ("set", 0, None), ("get", 0, None), ("set", 2, {"nested": {"k": [1.5, None, True]}}), ("get", 2, None), ("set", 0, b"\x00" * 240), ("get", 0, None),
]


def run(array_cls, smm) -> dict:
    arr = array_cls(smm, 4, 256)
    trace = []
    for op, index, value in SCRIPT:
        rec: dict = {"op": op, "index": index}
        try:
            if op == "set":
                rec["pickled_size"] = len(pickle.dumps(value))
                arr[index] = value
                rec["result"] = "ok"
            else:
                got = arr[index]
                rec["result"] = {"item": got.hex() if isinstance(got, bytes) else got, "bytes": isinstance(got, bytes)}
        except Exception as e:  # noqa: BLE001
            rec["result"] = {"raises": type(e).__name__, "message": str(e)}
        rec["max_actual_entry_size"] = arr.max_actual_entry_size()
        trace.append(rec)
    out = {"len": len(arr), "memory_size_at_least": 4 * 256, "memory_size": arr.get_memory_size(), "trace": trace, "ctor_errors": {}}
    for name, args in (("zero_entries", (0, 16)), ("zero_size", (4, 0))):
        try:
            array_cls(smm, *args)
            out["ctor_errors"][name] = None
        except Exception as e:  # noqa: BLE001
            out["ctor_errors"][name] = {"raises": type(e).__name__, "message": str(e)}
    return out


def main() -> None:
    sys.path.insert(0, "/root/reference")
    from pipelinerl.shared_memory_array import SharedMemoryArray

    with SharedMemoryManager() as smm:
        out = run(SharedMemoryArray, smm)
    values = [v.hex() if isinstance(v, bytes) else v for _, _, v in SCRIPT]
    (HERE / "array_trace.json").write_text(json.dumps({"script": [[op, i, v] for (op, i, _), v in zip(SCRIPT, values)], **out}, indent=1) + "\n")
    print("wrote array_trace.json:", len(out["trace"]), "operations")


if __name__ == "__main__":
    main()
