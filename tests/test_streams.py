"""Stream transports (CPU only): the files backend's wire format (one JSON object per line,
tensors as nested lists — reference streams.py:249-346), the binary shm backend, partition
writers, and that a batch survives both transports unchanged."""

import json
import threading

import numpy as np
import pytest
import torch

from oracle import preprocess as opre


@pytest.fixture()
def streams(tmp_path):
    from pipelinerl_amd import streams as s

    s.reset_streams_backend()
    yield s
    s.reset_streams_backend()


def _batch():
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.synthetic import make_entries

    data = opre.preprocess_chunk(make_entries(1, attempts=3, seq_length=24, vocab=50, seed=1, prompt_min=2, prompt_max=5), 2, True)
    d = opre.collate_packed(data, 2, 1)
    return PipelineBatchEncoding(**{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}), d


def _same(batch_kwargs, want):
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    got = PipelineBatchEncoding(**batch_kwargs)
    for k, v in want.items():
        g = getattr(got, k)
        if isinstance(v, np.ndarray):
            assert g.dtype == torch.from_numpy(v).dtype and np.array_equal(g.numpy(), v), k
        else:
            assert g == v, k


def test_backend_must_be_set_once(streams, tmp_path):
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="t")
    with pytest.raises(ValueError):
        streams.read_stream(spec)
    streams.set_streams_backend("files")
    with pytest.raises(ValueError):
        streams.set_streams_backend("files")
    streams.reset_streams_backend()
    # `redis` is the reference's own wire format and needs the redis client: without it the choice fails loudly (ImportError)
    # instead of being served by another transport
    try:
        import redis  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="redis"):
            streams.set_streams_backend("redis", host="localhost", port=6379)
    streams.reset_streams_backend()
    with pytest.raises(ValueError):
        streams.set_streams_backend("carrier-pigeon")


def test_files_wire_format_and_roundtrip(streams, tmp_path):
    streams.set_streams_backend("files")
    batch, want = _batch()
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="training_data", partition=3)
    assert str(spec) == "training_data/0/3"
    with streams.write_to_streams(spec) as w:
        w.write(batch)
        w.write({"kind": "samples_processed", "samples_processed": 7})
        with pytest.raises(ValueError):
            w.write({}, partition=0)
    path = tmp_path / "streams" / "training_data" / "0" / "3" / "0.jsonl"
    lines = path.read_text().splitlines()
    assert len(lines) == 2
    rec = json.loads(lines[0])
    # tensors are nested lists, scalars plain, optional tensors null — the reference's record layout
    assert rec["input_ids"] == want["input_ids"].tolist() and rec["seq_boundaries"] == want["seq_boundaries"].tolist()
    assert rec["is_packed"] is True and rec["pixel_values"] is None
    assert list(rec.keys())[:5] == ["input_ids", "attention_mask", "labels", "position_ids", "segment_ids"]
    with streams.read_stream(spec) as r:
        it = r.read()
        _same(next(it), want)
        assert next(it) == {"kind": "samples_processed", "samples_processed": 7}


def test_file_reader_tails_partial_lines(streams, tmp_path):
    streams.set_streams_backend("files")
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")
    d = streams.stream_dir(tmp_path, "actor", 0, 0)
    d.mkdir(parents=True)
    f = open(d / "0.jsonl", "w")
    f.write('{"a": 1}\n{"b"')
    f.flush()
    got = []

    def reader():
        rd = streams.FileStreamReader(spec, poll_delay=0.01)
        with rd:
            for rec in rd.read():
                got.append(rec)
                if len(got) == 2:
                    return

    t = threading.Thread(target=reader, daemon=True)
    t.start()
    import time

    time.sleep(0.1)
    assert got == [{"a": 1}]
    f.write(': 2}\n')
    f.flush()
    t.join(timeout=5)
    assert got == [{"a": 1}, {"b": 2}]


def test_partitioned_writer(streams, tmp_path):
    streams.set_streams_backend("files")
    rng = streams.StreamRangeSpec(exp_path=tmp_path, topic="training_data", partition_range=(0, 3))
    assert str(rng) == "training_data/0/0-3"
    with streams.write_to_streams(rng) as w:
        for i in range(4):
            w.write({"i": i})           # round robin: 0,1,2,0
        w.write({"i": 99}, partition=2)
        with pytest.raises(ValueError):
            w.write({}, partition=3)
    def lines(p):
        return [json.loads(x) for x in (tmp_path / "streams" / "training_data" / "0" / str(p) / "0.jsonl").read_text().splitlines()]
    assert lines(0) == [{"i": 0}, {"i": 3}] and lines(1) == [{"i": 1}] and lines(2) == [{"i": 2}, {"i": 99}]


def test_shm_backend_roundtrip(streams, tmp_path, libprl):
    streams.set_streams_backend("shm", segment_bytes=1 << 16, owner=True)
    batch, want = _batch()
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="training_data", partition=1)
    with streams.write_to_streams(spec) as w:
        w.write(batch)
        w.write({"kind": "training_done"})
        with streams.read_stream(spec) as r:
            it = r.read()
            _same(next(it), want)
            assert next(it) == {"kind": "training_done"}


def test_batch_codec_is_compact(libprl):
    from pipelinerl_amd import batch_codec
    from pipelinerl_amd.streams import _dumps

    batch, want = _batch()
    blob = batch_codec.encode_batch(batch)
    T = want["input_ids"].shape[1]
    assert len(blob) <= 68 * T + 4 * len(want["seq_boundaries"]) + 2048  # 5 x i64 + 7 x f32 per token + header
    assert len(_dumps(batch)) > len(blob) / 2  # the JSON form is of the same order or larger even on tiny batches
    _same(batch_codec.decode(blob), want)


def test_batch_codec_scalars_and_dtypes(libprl):
    """Round-3 advisor finding: a 0-dim CPU tensor has no byte view (`t.view(torch.uint8)` raises) and bool / bf16 / f16
    fields were encoded but not decodable - the failure surfaced in the CONSUMER.  Now: 0-dim tensors travel, the three
    dtypes are part of the format, anything else is refused at the producer."""
    import torch

    from pipelinerl_amd import batch_codec

    fields = [("scalar", torch.tensor(7, dtype=torch.int64)), ("flag", torch.tensor([True, False, True])),
              ("half", torch.arange(5, dtype=torch.float32).to(torch.bfloat16)), ("h16", torch.arange(3, dtype=torch.float16)),
              ("empty", torch.empty(0, dtype=torch.float32))]
    rec = batch_codec._frame(batch_codec.MAGIC_BATCH, {"model_version": 0}, fields)
    got = batch_codec.decode(rec)
    for name, t in fields:
        assert got[name].dtype == t.dtype and got[name].shape == t.shape and torch.equal(got[name], t), name
    with pytest.raises(TypeError, match="complex64"):
        batch_codec._frame(batch_codec.MAGIC_BATCH, {}, [("z", torch.zeros(2, dtype=torch.complex64))])


def test_rollouts_binary_record_roundtrip(streams, tmp_path, libprl):
    """The `actor` hop in binary: RaggedRollouts -> shm ring -> RaggedRollouts, field for field; and
    the text form written by the files backend is the reference's list-of-dicts group record."""
    from pipelinerl_amd import batch_codec
    from pipelinerl_amd.ragged import RaggedRollouts
    from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries

    rag, reasons = make_ragged(2, attempts=3, seq_length=30, vocab=40, seed=8, prompt_min=2, prompt_max=6, with_ref=True)
    got = batch_codec.decode(batch_codec.encode_rollouts(rag))
    assert isinstance(got, RaggedRollouts) and got.group_ids == rag.group_ids
    for name in ("tokens", "labels", "logprobs", "ref_logprobs", "seq_off", "lp_off", "reward", "group_index", "step_index",
                 "rollout_index", "model_version", "finished", "finish_code"):
        a, b = getattr(rag, name), getattr(got, name)
        assert a.dtype == b.dtype and torch.equal(a, b), name
    assert np.array_equal(got.host_seq_off, rag.host_seq_off)
    blob = batch_codec.encode_rollouts(rag)
    assert len(blob) < 20 * rag.n_tokens + 4096  # ~16 B/token + header
    streams.set_streams_backend("files")
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")
    with streams.write_to_streams(spec) as w:
        w.write(rag)
    with streams.read_stream(spec) as r:
        rec = next(r.read())
    want = ragged_to_entries(rag)
    assert [e["input_ids"] for e in rec] == [e["input_ids"] for e in want]
    assert [e["metadata"] for e in rec] == [e["metadata"] for e in want]


# ---- interop with files written / read by the reference's own `files` backend -------------------
# tests/golden/streams_files.json comes from tests/golden/make_streams_golden.py, which runs
# pipelinerl/streams.py (reference :249-423) itself with stand-ins for the two missing imports.


def _golden_streams():
    from helpers import GOLDEN

    return json.loads((GOLDEN / "streams_files.json").read_text())


def _replay_scenario(s, exp):
    """The same writes as make_streams_golden.scenario, through this package's API."""
    from pydantic import BaseModel

    class Msg(BaseModel):
        kind: str = "samples_processed"
        samples_processed: int = 0

    class WithTensor(BaseModel):
        model_config = {"arbitrary_types_allowed": True}
        input_ids: torch.Tensor
        rewards: torch.Tensor
        model_version: int = 3
        is_packed: bool = True

    with s.write_to_streams(s.SingleStreamSpec(exp_path=exp, topic="actor")) as w:
        w.write([{"input_ids": [1, 2, 3], "labels": [-100, 2, 3], "reward": 1.0, "logprobs": [-0.5, -0.25], "metadata": {"group_id": "g0"}}])
        w.write([{"input_ids": [4], "labels": [4], "reward": 0.0, "logprobs": [-1.5], "metadata": {"group_id": "g1", "nested": [1, {"a": None}]}}])
    rng = s.StreamRangeSpec(exp_path=exp, topic="training_data", partition_range=(0, 3))
    with s.write_to_streams(rng) as w:
        for i in range(5):
            w.write({"i": i})
        w.write({"i": "explicit"}, partition=2)
        w.write(WithTensor(input_ids=torch.tensor([[5, 6, 7]]), rewards=torch.tensor([[0.5, 0.0, 1.0]])), partition=1)
    spec = s.SingleStreamSpec(exp_path=exp, topic="stats", instance=2, partition=1)
    with s.write_to_streams(spec) as w:
        w.write(Msg(samples_processed=8))
    with s.write_to_streams(spec) as w:
        w.write(Msg(samples_processed=16))
    spec_w = s.SingleStreamSpec(exp_path=exp, topic="weight_update_request")
    with s.write_to_streams(spec_w) as w:
        w.write({"version": 1})
    with s.write_to_streams(spec_w, mode="w") as w:
        w.write({"version": 2, "np": np.arange(3)})
    return str(spec), str(rng)


def test_files_backend_writes_what_the_reference_writes(streams, tmp_path):
    """Same directory tree, same number of lines per file, same parsed record per line."""
    g = _golden_streams()
    streams.set_streams_backend("files")
    str_single, str_range = _replay_scenario(streams, tmp_path)
    assert str_single == g["str_single"] and str_range == g["str_range"]
    mine = {str(p.relative_to(tmp_path)): p.read_text() for p in sorted(tmp_path.rglob("*")) if p.is_file()}
    assert sorted(mine) == sorted(g["files"])
    for rel, want_text in g["files"].items():
        got_lines, want_lines = mine[rel].split("\n"), want_text.split("\n")
        assert got_lines[-1] == "" and len(got_lines) == len(want_lines), rel  # newline-terminated, one record per line
        assert [json.loads(l) for l in got_lines[:-1]] == [json.loads(l) for l in want_lines[:-1]], rel


def test_files_backend_reads_reference_written_files(streams, tmp_path):
    """Files produced by the reference are consumed record for record (debug.streams_from replay),
    and the reader yields exactly what the reference's own reader yielded."""
    g = _golden_streams()
    for rel, text in g["files"].items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    streams.set_streams_backend("files")
    for rel, want in g["records"].items():
        _, topic, instance, partition, _ = rel.split("/")
        spec = streams.SingleStreamSpec(exp_path=tmp_path, topic=topic, instance=int(instance), partition=int(partition))
        with streams.read_stream(spec) as r:
            it = r.read()
            got = [next(it) for _ in range(len(want))]
        assert got == want, rel


def test_shm_backend_jsonl_mirror_allows_replay(streams, tmp_path):
    """shm transport with `mirror_jsonl`: the binary ring feeds the live reader, the JSONL mirror
    under <exp>/streams/... replays the same records later through the files backend."""
    from pipelinerl_amd.ragged import RaggedRollouts
    from pipelinerl_amd.synthetic import make_ragged

    rag, _ = make_ragged(2, attempts=3, seq_length=40, vocab=60, seed=4, prompt_min=3, prompt_max=8, with_ref=True)
    batch, want_batch = _batch()
    streams.set_streams_backend("shm", segment_bytes=1 << 20, mirror_jsonl=["actor", "training_data"], owner=True)
    a = streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")
    t = streams.SingleStreamSpec(exp_path=tmp_path, topic="training_data")
    s = streams.SingleStreamSpec(exp_path=tmp_path, topic="stats")
    with streams.write_to_streams(a) as wa, streams.write_to_streams(t) as wt, streams.write_to_streams(s) as ws:
        wa.write(rag)
        wt.write(batch)
        ws.write({"x": 1})
        with streams.read_stream(a) as r:
            live = next(iter(r.read()))
        assert isinstance(live, RaggedRollouts) and torch.equal(live.tokens, rag.tokens)
    assert not (tmp_path / "streams" / "stats").exists()  # topic not mirrored
    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    with streams.read_stream(a) as r:
        entries = next(iter(r.read()))
    back = RaggedRollouts.from_entries(entries)
    for name in ("tokens", "labels", "logprobs", "ref_logprobs", "seq_off", "lp_off", "reward", "step_index", "rollout_index",
                 "model_version", "finished", "finish_code"):
        assert torch.equal(getattr(back, name), getattr(rag, name)), name
    assert [back.group_ids[i] for i in back.host_group_index] == [rag.group_ids[i] for i in rag.host_group_index]
    with streams.read_stream(t) as r:
        _same(next(iter(r.read())), want_batch)


def test_packed_step_record_recipe_equals_the_generic_encoding():
    """`PackedStep.describe_record(j)` (arithmetic on the block's geometry, what the native publisher is handed) gathers to exactly
    the bytes `batch_codec.encode_batch(packed[j])` produces - header, alignment gaps, every column, seq_boundaries."""
    import numpy as np
    import torch

    from pipelinerl_amd import batch_codec
    from pipelinerl_amd.finetune.data import PackedStep, _alloc_outputs

    rng = np.random.default_rng(3)
    lens = [5, 9, 3, 12, 7, 1, 4]
    pk_dst = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    mb_off = np.array([0, 2, 3, 7], dtype=np.int64)  # three micro-batches: 2, 1 and 4 sequences
    total = int(pk_dst[-1])
    out = _alloc_outputs(total, torch.device("cpu"), packed=True)
    out["__block__"].copy_(torch.from_numpy(rng.integers(0, 255, out["__block__"].numel(), dtype=np.uint8)))
    packed = PackedStep(out, pk_dst, mb_off, np.array([3, 7, 7], dtype=np.int64), np.array([0, 1, 0], dtype=np.int64))
    block = packed.block.numpy().tobytes()
    for j in range(3):
        inline = bytearray(b"junk")  # recipes append to what is there already
        nbytes, pieces = packed.describe_record(j, inline)
        rec = bytearray(nbytes)
        for kind, src, off, nb in pieces:
            rec[off:off + nb] = (block if kind == 0 else bytes(inline))[src:src + nb]
        want = batch_codec.encode_batch(packed[j])
        assert bytes(rec) == bytes(want), j
        d = batch_codec.decode(rec)
        assert d["model_version"] == [3, 7, 7][j] and d["padding"] == [0, 1, 0][j] and d["seq_boundaries"].tolist()[-1] == d["input_ids"].shape[1]
    # the generic recipe (tensor views: sequence-parallel slices, sentinels) agrees with it where both apply
    inline_a, inline_b = bytearray(), bytearray()
    na, pa = packed.describe_record(1, inline_a)
    dev_like = packed[1]
    nb_, pb = batch_codec.describe_batch(dev_like, packed.block.data_ptr(), packed.block.numel(), inline_b)
    assert na == nb_


# ---------------------------------------------------------------------------------------------
# redis backend against an in-process stand-in of the client calls it makes (the image has neither client nor server)
# ---------------------------------------------------------------------------------------------


class _FakeRedisServer:
    """The stream commands of one server: XADD (auto ids `<ms>-<seq>`, field names and values come back as bytes, like redis-py
    without decode_responses), XREVRANGE count=1, XREAD {name: last_id} count=1 block=ms."""

    def __init__(self):
        self.streams: dict[str, list] = {}
        self.cond = threading.Condition()
        self.refuse_first_pings = 0

    def client_module(self):
        import types

        server = self

        class ConnectionError(Exception):  # noqa: A001 - the name redis-py uses
            pass

        class TimeoutError(Exception):  # noqa: A001
            pass

        class Redis:
            def __init__(self, host="localhost", port=6379):
                self.host, self.port, self.closed = host, port, False

            def ping(self):
                if server.refuse_first_pings > 0:
                    server.refuse_first_pings -= 1
                    raise ConnectionError("connection refused")
                return True

            def close(self):
                self.closed = True

            def xadd(self, name, fields, maxlen=None, approximate=True):
                assert maxlen == 1000000 and approximate is True  # the reference's retention (streams.py:157)
                with server.cond:
                    entries = server.streams.setdefault(name, [])
                    eid = (len(entries) + 1, 0)
                    enc = lambda v: v if isinstance(v, bytes) else str(v).encode()  # noqa: E731
                    entries.append((eid, {k.encode(): enc(v) for k, v in fields.items()}))
                    server.cond.notify_all()
                return f"{eid[0]}-{eid[1]}".encode()

            def xrevrange(self, name, count=None):
                with server.cond:
                    entries = server.streams.get(name, [])
                    return [(f"{e[0][0]}-{e[0][1]}".encode(), e[1]) for e in reversed(entries)][:count]

            def xread(self, streams, count=None, block=None):
                (name, last), = streams.items()
                if isinstance(last, bytes):
                    last = tuple(int(x) for x in last.decode().split("-"))
                elif last == 0:
                    last = (0, 0)
                with server.cond:
                    def newer():
                        return [e for e in server.streams.get(name, []) if e[0] > last]
                    if not newer() and block:
                        server.cond.wait(block / 1000.0)
                    got = newer()[:count]
                return [[name.encode(), [(f"{e[0][0]}-{e[0][1]}".encode(), e[1]) for e in got]]] if got else []

        mod = types.ModuleType("redis")
        mod.Redis, mod.ConnectionError = Redis, ConnectionError
        mod.exceptions = types.SimpleNamespace(TimeoutError=TimeoutError, ConnectionError=ConnectionError)
        return mod


@pytest.fixture()
def fake_redis(monkeypatch, streams):
    import sys

    server = _FakeRedisServer()
    monkeypatch.setitem(sys.modules, "redis", server.client_module())
    monkeypatch.setattr(streams, "_REDIS_RETRY_DELAY", 0.01)
    streams.set_streams_backend("redis", host="redis.example", port=6380)
    return server


def test_redis_wire_format_is_the_references(streams, fake_redis, tmp_path):
    """Stream name `topic/instance/partition`, fields {index, data}, data = pickle of the record's plain form (streams.py:120-158)."""
    import pickle

    from pipelinerl_amd.finetune_loop import SamplesProcessed

    batch, want = _batch()
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="training_data", partition=3)
    with streams.write_to_streams(spec) as w:
        assert isinstance(w, streams.RedisStreamWriter)
        w.write(batch)
        w.write(SamplesProcessed(samples_processed=7))
        with pytest.raises(ValueError):
            w.write({}, partition=0)
    entries = fake_redis.streams["training_data/0/3"]
    assert [int(e[1][b"index"]) for e in entries] == [0, 1] and set(entries[0][1]) == {b"index", b"data"}
    first, second = (pickle.loads(e[1][b"data"]) for e in entries)
    assert isinstance(first, dict) and isinstance(first["input_ids"], torch.Tensor)  # model_dump(): tensors stay tensors in the pickle
    _same(first, want)
    assert second["kind"] == "samples_processed" and second["samples_processed"] == 7
    with streams.read_stream(spec) as r:
        it = r.read()
        _same(next(it), want)
        assert next(it)["samples_processed"] == 7


def test_redis_append_continues_the_index_and_w_refuses_existing_data(streams, fake_redis, tmp_path):
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="trainer")
    for k in range(3):  # the trainer topic is reopened for every weight update (finetune_loop.py:244)
        with streams.write_to_streams(spec) as w:
            w.write({"k": k})
    assert [int(e[1][b"index"]) for e in fake_redis.streams["trainer/0/0"]] == [0, 1, 2]
    with pytest.raises(ValueError, match="already exists"):
        streams.write_to_streams(spec, mode="w")
    with pytest.raises(ValueError, match="Invalid mode"):
        streams.write_to_streams(spec, mode="x")
    fresh = streams.SingleStreamSpec(exp_path=tmp_path, topic="fresh")
    with streams.write_to_streams(fresh, mode="w") as w:
        w.write({"a": 1})
    # two independent readers each see every record from the first one (fan-out: TrainerState in three processes)
    for _ in range(2):
        with streams.read_stream(spec) as r:
            it = r.read()
            assert [next(it)["k"] for _ in range(3)] == [0, 1, 2]


def test_redis_reader_blocks_for_new_entries_and_checks_the_running_index(streams, fake_redis, tmp_path):
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")
    got = []

    def consume():
        with streams.read_stream(spec) as r:
            for rec in r.read():
                got.append(rec)
                if len(got) == 2:
                    return

    t = threading.Thread(target=consume)
    t.start()
    with streams.write_to_streams(spec) as w:  # the reader polls an empty / absent stream until entries arrive
        w.write([{"reward": 1.0}])
        w.write([{"reward": 0.0}])
    t.join(10)
    assert not t.is_alive() and got == [[{"reward": 1.0}], [{"reward": 0.0}]]
    # a stream that lost its head (maxlen trimming) or has a foreign writer: the index check raises, as in the reference (:185-186)
    fake_redis.streams["actor/0/0"].pop(0)
    with streams.read_stream(spec) as r, pytest.raises(ValueError, match="Index mismatch"):
        next(r.read())


def test_redis_partitioned_writer_and_connection_retry(streams, fake_redis, tmp_path):
    fake_redis.refuse_first_pings = 2  # the server is not up yet: unlimited retries (:106-117)
    rng = streams.StreamRangeSpec(exp_path=tmp_path, topic="training_data", partition_range=(0, 3))
    with streams.write_to_streams(rng) as w:
        for k in range(4):
            w.write({"k": k})          # round robin
        w.write({"k": 99}, partition=2)
        with pytest.raises(ValueError):
            w.write({}, partition=3)
    assert fake_redis.refuse_first_pings == 0
    import pickle

    by_part = {p: [pickle.loads(e[1][b"data"])["k"] for e in fake_redis.streams[f"training_data/0/{p}"]] for p in range(3)}
    assert by_part == {0: [0, 3], 1: [1], 2: [2, 99]}


def test_redis_options_are_host_and_port_only(streams, monkeypatch):
    import sys
    import types

    monkeypatch.setitem(sys.modules, "redis", types.ModuleType("redis"))
    with pytest.raises(ValueError, match="host and port"):
        streams.set_streams_backend("redis", host="h", segment_bytes=1)


def test_reference_class_names_of_the_partitioned_writers(streams, fake_redis, tmp_path):
    """`RoundRobinFileStreamWriter` / `RoundRobinRedisStreamWriter` / `RedisConfig` / `connect_to_redis` exist under the reference's names
    (streams.py:25-30, 106-117, 195-232, 349-384)."""
    rng = streams.StreamRangeSpec(exp_path=tmp_path, topic="t", partition_range=(0, 2))
    with streams.RoundRobinRedisStreamWriter(rng) as w:
        for k in range(3):
            w.write({"k": k})
    assert [len(fake_redis.streams[f"t/0/{p}"]) for p in range(2)] == [2, 1]
    with streams.RoundRobinFileStreamWriter(rng, mode="w") as w:
        w.write({"k": 0}, partition=1)
    assert (tmp_path / "streams" / "t" / "0" / "1" / "0.jsonl").read_text() == '{"k":0}\n'
    client = streams.connect_to_redis(streams.RedisConfig(host="other.example", port=1234))
    assert (client.host, client.port) == ("other.example", 1234) and streams._backend_options["host"] == "redis.example"
    assert streams.connect_to_redis().host == "redis.example"
