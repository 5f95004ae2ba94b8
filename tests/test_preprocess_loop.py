"""The lossy / elastic half of the preprocessor loop against traces of the reference's own code
(tests/golden/make_preprocess_loop_golden.py cuts `run_dataset_loader`, the ring block, the throughput
aggregator, the stats block and `replace_oov_tokens_with_the` out of pipelinerl/preprocess.py and executes
them with recording stubs).  CPU only."""

import json
import queue
import types
from collections import deque

import pytest

from helpers import GOLDEN

G = json.loads((GOLDEN / "preprocess_loop.json").read_text())


class StopTrace(BaseException):
    pass


@pytest.mark.parametrize("case", G["dataset_loader"], ids=lambda c: "q{raw_queue_size}_n{chunk_n_groups}_pop{pop_old_data}_c{consume_every}".format(**c["params"]))
def test_chunk_loader_drops_the_oldest_chunk_like_the_reference(case):
    from pipelinerl_amd.preprocess import ChunkLoader

    p, groups, ops = case["params"], case["groups"], []

    def cid(item):
        return "error" if isinstance(item, Exception) else [e["uid"] for g in item for e in g]

    class TraceQueue(queue.Queue):
        def put(self, item, block=True, timeout=None):
            if block and self.full():
                ops.append(["would_block", cid(item)])
                raise StopTrace()
            try:
                queue.Queue.put(self, item, block, timeout)
                ops.append(["put", cid(item)])
            except queue.Full:
                ops.append(["full", cid(item)])
                raise

        def get_nowait(self):
            item = queue.Queue.get(self, block=False)
            ops.append(["drop", cid(item)])
            return item

    q = TraceQueue(p["raw_queue_size"])
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
                if p["consume_every"] and fed["n"] % p["consume_every"] == 0 and not q.empty():
                    ops.append(["consume", cid(queue.Queue.get_nowait(q))])
                yield g
            raise StopTrace()

    loader = ChunkLoader(q, None, p["attempts"], p["chunk_n_groups"], p["pop_old_data"], reader_factory=lambda spec: Reader())
    with pytest.raises(StopTrace):
        loader.run()
    left = []
    while not q.empty():
        left.append(cid(queue.Queue.get_nowait(q)))
    assert ops == case["ops"] and left == case["left"]
    assert loader.old_and_dropped == sum(1 for o in case["ops"] if o[0] == "drop")


@pytest.mark.parametrize("case", G["ring"], ids=lambda c: f"maxlen{c['maxlen']}_pop{c['pop_old_data']}")
def test_processed_ring_admission_matches_the_reference(case):
    from pipelinerl_amd.preprocess import ProcessedRing

    updates = []
    ring = ProcessedRing(case["maxlen"], case["pop_old_data"], stats=types.SimpleNamespace(update=lambda c: updates.append(list(c))),
                         length_of=lambda e: e["length"], version_of=lambda e: e["model_version"])
    buffer: deque = deque()
    uid = 0
    for step in case["steps"]:
        for _ in range(step["arrived"]):
            buffer.append({"uid": uid, **case["entry_attrs"][uid]})
            uid += 1
        before = len(updates)
        ring.admit(buffer)
        assert [e["uid"] for e in ring.entries] == step["ring"]
        assert [e["uid"] for e in buffer] == step["buffer"]
        assert ring.popped == step["popped"] and ring.max_model_version == step["max_model_version"]
        assert updates[before:] == step["stat_updates"]
        for _ in range(step["take"]):
            if ring.entries:
                ring.entries.popleft()


def test_sliding_window_aggregator_matches_the_reference():
    from pipelinerl_amd.preprocess import SlidingWindowAggregator

    g = G["aggregator"]
    clock = {"t": 0.0}
    agg = SlidingWindowAggregator(g["window_size"], clock=lambda: clock["t"])
    for step in g["trace"]:
        clock["t"] = step["t"]
        agg.update(step["counts"])
        assert agg.has_enough_data() == step["enough"]
        got = agg.get_stats()
        assert got == pytest.approx(step["stats"], rel=1e-12)
    assert SlidingWindowAggregator(2).get_stats() == g["empty_stats"]


@pytest.mark.parametrize("case", G["stats_block"], ids=lambda c: "pub{published_samples}_last{last_published_samples}_{debug_mode}_{batch_done}".format(**c["inputs"]))
def test_preprocessor_stats_record_and_trigger(case):
    from pipelinerl_amd.preprocess import preprocessor_stats_record, should_write_stats

    i = case["inputs"]
    fire = should_write_stats(i["published_samples"], i["last_published_samples"], i["debug_mode"], i["batch_done"], i["log_every_n_samples"])
    assert fire == bool(case["written"])
    if fire:
        agg = types.SimpleNamespace(has_enough_data=lambda: i["enough"], get_stats=lambda: {"samples_per_second": 12.5, "tokens_per_second": 999.0})
        rec = preprocessor_stats_record(i["published_samples"], 7, raw_queue_chunks=5, output_queue_chunks=3, chunk_n_groups=2, attempts=8,
                                        num_filtered_out=4, total_filtered_out=11, aggregator=agg)
        assert rec == case["written"][0] and list(rec) == list(case["written"][0])


def test_oov_patch_host_front_end_matches_the_reference():
    from pipelinerl_amd.preprocess import replace_oov_tokens_with_the

    g = G["oov"]
    vocab = {f"t{i}": i for i in g["vocab_ids"] if i != g["the_token_id"]}
    vocab["the"] = g["the_token_id"]
    tok = types.SimpleNamespace(get_vocab=lambda: dict(vocab))
    data = json.loads(json.dumps(g["data"]))
    out = replace_oov_tokens_with_the(data, tok)
    assert [e["input_ids"] for e in out] == g["patched_input_ids"]
    assert [e["labels"] for e in out] == g["labels_after"]  # labels are not patched (nor by the reference)


def test_pop_old_data_rule():
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.preprocess import PreprocessorConfig

    base = dict(exp_path="x", num_trainers=1, train_batch_size=1, gradient_accumulation_passes=1, seq_length=8, attempts=1, rl=RLConfig(), eos_token_id=2)
    assert PreprocessorConfig(**base).drops_old_data  # conf/base.yaml: pop_old_data true, max_lag null, no debug mode
    assert not PreprocessorConfig(**base, max_lag=4).drops_old_data
    assert not PreprocessorConfig(**base, debug_mode="preprocessor").drops_old_data
    assert not PreprocessorConfig(**base, pop_old_data=False).drops_old_data
