"""Golden traces of the lossy / elastic half of the reference's preprocessor loop, produced by EXECUTING
the reference's own code (`pipelinerl/preprocess.py` cannot be imported here: litellm / omegaconf / redis
are absent).  Each piece is cut from the source at generation time and `exec`ed with recording stubs;
nothing of the reference is stored in this repository - only the traces (tests/golden/preprocess_loop.json).

  * `run_dataset_loader` (:190-228)       chunking of groups + drop-oldest on the bounded raw-chunk queue
  * the ring block (:572-585)             buffer -> processed ring (deque(maxlen)) with `pop_old_data`
  * `SlidingWindowAggregator` (:239-282)  samples / tokens per second over the last N chunks
  * the stats block (:664-694)            what goes to the `preprocessor_stats` stream, and when
  * `replace_oov_tokens_with_the` (:107-141)

    python tests/golden/make_preprocess_loop_golden.py
"""

from __future__ import annotations

import json
import queue
import textwrap
import types
from collections import defaultdict, deque
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = Path("/root/reference/pipelinerl/preprocess.py")
LINES = SRC.read_text().splitlines()


def top_level(name: str, kind: str = "def") -> str:
    start = next(i for i, l in enumerate(LINES) if l.startswith(f"{kind} {name}"))
    end = start + 1
    while end < len(LINES) and (not LINES[end].strip() or LINES[end].startswith((" ", "\t", ")"))):
        end += 1
    return "\n".join(LINES[start:end])


def block_starting(prefix: str) -> str:
    """The statement that starts with `prefix` (stripped) inside run_preprocessing_loop, with its body, dedented."""
    start = next(i for i, l in enumerate(LINES) if l.strip().startswith(prefix))
    indent = len(LINES[start]) - len(LINES[start].lstrip())
    end = start + 1
    # a multi-line condition: continue until the line that ends the header
    while not LINES[end - 1].rstrip().endswith(":"):
        end += 1
    while end < len(LINES) and (not LINES[end].strip() or len(LINES[end]) - len(LINES[end].lstrip()) > indent):
        end += 1
    return textwrap.dedent("\n".join(LINES[start:end]))


QUIET = types.SimpleNamespace(debug=lambda *a, **k: None, info=lambda *a, **k: None, warning=lambda *a, **k: None, error=lambda *a, **k: None)


class StopTrace(BaseException):
    pass


# ------------------------------------------------------------------------------------------------
def gen_dataset_loader():
    ns = {"queue": queue, "Empty": queue.Empty, "Queue": queue.Queue, "SingleStreamSpec": object, "logger": QUIET, "defaultdict": defaultdict}
    exec(compile(top_level("_check_group_sizes"), "ref_check", "exec"), ns)
    exec(compile(top_level("run_dataset_loader"), "ref_loader", "exec"), ns)
    cases = []
    rng = np.random.default_rng(3)
    for qsize, chunk_n, n_groups, attempts, pop_old, consume_every, bad_at in [
        (3, 2, 20, 2, True, 0, None), (3, 2, 20, 2, True, 4, None), (2, 1, 9, 3, True, 0, None), (4, 3, 30, 2, False, 2, None),
        (8, 2, 12, 2, True, 0, None), (3, 2, 14, 2, True, 0, 9),
    ]:
        groups = []
        for g in range(n_groups):
            size = attempts if g != bad_at else attempts - 1
            groups.append([{"group_id": f"g{g}", "metadata": {"rollout_index": r}, "uid": g * 10 + r} for r in range(size)])
        ops = []

        class TraceQueue(queue.Queue):
            def put(self, item, block=True, timeout=None):  # put_nowait() arrives here with block=False
                if block and self.full():
                    ops.append(["would_block", _cid(item)])
                    raise StopTrace()
                try:
                    queue.Queue.put(self, item, block, timeout)
                    ops.append(["put", _cid(item)])
                except queue.Full:
                    ops.append(["full", _cid(item)])
                    raise

            def get_nowait(self):
                item = queue.Queue.get(self, block=False)
                ops.append(["drop", _cid(item)])
                return item

        def _cid(item):
            return "error" if isinstance(item, Exception) else [e["uid"] for e in item]

        q = TraceQueue(qsize)
        fed = {"n": 0}

        class Reader:
            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

            def read(self):
                while fed["n"] < len(groups):
                    g = groups[fed["n"]]
                    fed["n"] += 1
                    if consume_every and fed["n"] % consume_every == 0 and not q.empty():
                        item = queue.Queue.get_nowait(q)  # the main loop takes a chunk now and then
                        ops.append(["consume", _cid(item)])
                    yield g
                raise StopTrace()

        ns["read_stream"] = lambda spec: Reader()
        try:
            ns["run_dataset_loader"](q, None, attempts, chunk_n, pop_old)
        except StopTrace:
            pass
        left = []
        while not q.empty():
            left.append(_cid(queue.Queue.get_nowait(q)))
        cases.append({"params": dict(raw_queue_size=qsize, chunk_n_groups=chunk_n, attempts=attempts, pop_old_data=pop_old, consume_every=consume_every),
                      "groups": groups, "ops": ops, "left": left})
    return cases


# ------------------------------------------------------------------------------------------------
def gen_ring():
    code = compile(block_starting("while len(buffer) > 0:"), "ref_ring", "exec")
    cases = []
    rng = np.random.default_rng(4)
    for maxlen, pop_old, arrivals, takes in [(4, True, [3, 3, 5, 1], [0, 1, 2, 0]), (4, False, [3, 3, 5, 1], [0, 1, 2, 4]), (8, True, [10, 2, 9], [3, 0, 8])]:
        updates = []
        ns = dict(buffer=deque(), processed_entries_queue=deque(maxlen=maxlen), pop_old_data=pop_old, processed_entries_queue_popped_data=0,
                  last_time_notice=0, logger=QUIET, max_model_version=None,
                  stats_aggregator=types.SimpleNamespace(update=lambda counts: updates.append(list(counts))))
        uid = 0
        steps = []
        for n_new, n_take in zip(arrivals, takes):
            for _ in range(n_new):
                ns["buffer"].append({"uid": uid, "input_ids": [0] * int(rng.integers(1, 30)), "model_version": int(rng.integers(0, 5))})
                uid += 1
            before = len(updates)
            exec(code, ns)
            steps.append({"arrived": n_new, "ring": [e["uid"] for e in ns["processed_entries_queue"]], "buffer": [e["uid"] for e in ns["buffer"]],
                          "popped": ns["processed_entries_queue_popped_data"], "max_model_version": ns["max_model_version"],
                          "stat_updates": updates[before:], "take": n_take})
            for _ in range(n_take):  # the scheduler consumes from the head of the ring
                if ns["processed_entries_queue"]:
                    ns["processed_entries_queue"].popleft()
        cases.append({"maxlen": maxlen, "pop_old_data": pop_old, "entries": uid, "steps": steps,
                      "lengths": None})
        # lengths / versions of every entry, for the replay
        rng2 = np.random.default_rng(4)
    # regenerate with recorded entry attributes (the rng stream above is consumed in order: replay it)
    rng = np.random.default_rng(4)
    for c in cases:
        ents = []
        for _ in range(c["entries"]):
            ents.append({"length": int(rng.integers(1, 30)), "model_version": int(rng.integers(0, 5))})
        c["entry_attrs"] = ents
        del c["lengths"]
    return cases


# ------------------------------------------------------------------------------------------------
def gen_aggregator():
    clock = {"t": 100.0}
    fake_time = types.SimpleNamespace(time=lambda: clock["t"])

    class BaseModel:  # stands in for pydantic.BaseModel in the dataclass-like SlidingWindowData
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

    ns = {"time": fake_time}
    # SlidingWindowData is a pydantic model with two list fields: a plain namespace serves the aggregator code
    ns["SlidingWindowData"] = lambda: types.SimpleNamespace(tokens_window=[], timestamps=[])
    exec(compile(top_level("SlidingWindowAggregator", "class"), "ref_agg", "exec"), ns)
    agg = ns["SlidingWindowAggregator"](window_size=3)
    trace = []
    rng = np.random.default_rng(5)
    for step in range(7):
        clock["t"] += float(rng.uniform(0.1, 2.0)) if step != 3 else 0.0
        counts = rng.integers(1, 100, size=int(rng.integers(1, 6))).tolist()
        agg.update(counts)
        trace.append({"t": clock["t"], "counts": counts, "enough": agg.has_enough_data(), "stats": agg.get_stats()})
    empty = ns["SlidingWindowAggregator"](window_size=2)
    return {"window_size": 3, "trace": trace, "empty_stats": empty.get_stats()}


# ------------------------------------------------------------------------------------------------
def gen_stats_block():
    code = compile(_stats_block_source(), "ref_stats", "exec")
    cases = []
    for published, last, debug_mode, batch_done, log_every, enough in [
        (64, 0, None, True, 128, True), (64, 0, None, False, 128, True), (300, 100, None, False, 128, False), (10, 10, "x", True, 1, True),
        (20, 10, "preprocessor", False, 128, False),
    ]:
        written = []
        ns = dict(
            published_samples=published, last_published_samples=last, batch_done=batch_done, max_model_version=7,
            cfg=types.SimpleNamespace(debug=types.SimpleNamespace(mode=debug_mode), attempts=8,
                                      preprocess=types.SimpleNamespace(log_every_n_samples=log_every, chunk_n_groups=2)),
            output_queue=types.SimpleNamespace(qsize=lambda: 3, max_actual_entry_size=lambda: 4242),
            raw_chunk_queue=types.SimpleNamespace(qsize=lambda: 5), num_filtered_out=4, total_filtered_out=11,
            stats_aggregator=types.SimpleNamespace(has_enough_data=lambda: enough, get_stats=lambda: {"samples_per_second": 12.5, "tokens_per_second": 999.0}),
            wandb_run=None, stats_writer=types.SimpleNamespace(write=lambda s: written.append(dict(s))),
            time=types.SimpleNamespace(time=lambda: 50.0), start_processing=40.0, fetching_took=1.0, writing_took=2.0, logger=QUIET,
            output_stream="training_data/0/0-1",
        )
        exec(code, ns)
        cases.append({"inputs": dict(published_samples=published, last_published_samples=last, debug_mode=debug_mode, batch_done=batch_done,
                                     log_every_n_samples=log_every, enough=enough),
                      "written": written, "after": {k: ns[k] for k in ("last_published_samples", "num_filtered_out", "fetching_took", "writing_took")}})
    return cases


def _stats_block_source() -> str:
    start = next(i for i, l in enumerate(LINES) if l.strip() == "if (" and "published_samples > last_published_samples" in LINES[i + 1])
    indent = len(LINES[start]) - len(LINES[start].lstrip())
    end = start + 1
    while not LINES[end - 1].strip() == "):":
        end += 1
    while end < len(LINES) and (not LINES[end].strip() or len(LINES[end]) - len(LINES[end].lstrip()) > indent):
        end += 1
    return textwrap.dedent("\n".join(LINES[start:end]))


# ------------------------------------------------------------------------------------------------
def gen_oov():
    ns = {"logger": QUIET, "transformers": types.SimpleNamespace(PreTrainedTokenizerBase=object)}
    exec(compile(top_level("replace_oov_tokens_with_the"), "ref_oov", "exec"), ns)
    vocab = {f"t{i}": i for i in range(0, 50) if i not in (7, 13)}  # ids 7 and 13 are holes in the vocabulary
    vocab["the"] = 21
    del vocab["t21"]
    tok = types.SimpleNamespace(get_vocab=lambda: dict(vocab))
    rng = np.random.default_rng(6)
    data = []
    for i in range(6):
        n = int(rng.integers(3, 12))
        ids = rng.integers(0, 60, size=n).tolist()  # >= 50, 7 and 13 are out of vocabulary
        data.append({"input_ids": ids, "labels": [-100] * 2 + ids[2:], "logprobs": [-0.1] * (n - 2)})
    before = json.loads(json.dumps(data))
    out = ns["replace_oov_tokens_with_the"](data, tok)
    return {"vocab_ids": sorted(vocab.values()), "the_token_id": 21, "data": before, "patched_input_ids": [e["input_ids"] for e in out],
            "labels_after": [e["labels"] for e in out]}


def main():
    out = {"dataset_loader": gen_dataset_loader(), "ring": gen_ring(), "aggregator": gen_aggregator(), "stats_block": gen_stats_block(), "oov": gen_oov()}
    (HERE / "preprocess_loop.json").write_text(json.dumps(out))
    for k, v in out.items():
        print(k, len(v) if isinstance(v, list) else "ok")


if __name__ == "__main__":
    main()
