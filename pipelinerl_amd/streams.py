"""Topic streams between the pipeline stages (drop-in for reference pipelinerl/streams.py).

Public API as in the reference (:33-69, :390-423): `set_streams_backend`, `SingleStreamSpec`,
`StreamRangeSpec`, `read_stream(spec)` -> context manager with `.read()` iterator,
`write_to_streams(spec, mode)` -> context manager with `.write(data, partition=None)`.

Backends
  files  the reference's on-disk layout and wire format, kept structurally identical so recorded runs can
         be replayed (same keys, nesting and line framing; numbers are printed by the stdlib
         encoder instead of orjson, so files are parse-compatible, not byte-identical) (`debug.streams_from`): `<exp>/streams/<topic>/<instance>/<partition>/0.jsonl`,
         one JSON object per line, tensors as nested lists, flush per record (:238-278).  The
         reader tails the file and never sees EOF (:281-346).
  shm    MI355X-native transport: one `prl_log` (csrc/prl_log.cpp) per (topic, instance, partition) - an
         append-only record LOG in POSIX shared memory with the reference backends' semantics: every
         reader sees every record from the first one, any number of readers per topic (`TrainerState`
         follows the trainer topic in the actor, the preprocessor and the launcher at once,
         state.py:36-48), the writer never waits for a reader, a writer that is closed and reopened
         appends to the same stream (finetune_loop.py:244), mode "w" starts it over.  A
         `PipelineBatchEncoding` travels as a binary SoA record (header + raw int64 / fp32 buffers, see
         `batch_codec`) instead of ~114 bytes of JSON text per token, and readers park on a futex
         instead of polling every 100 ms.  The bulk topics (`trim_topics`, default `training_data` and
         `actor`) drop segments every registered reader has consumed; everything else is retained
         like a file.  Non-batch records (trainer messages, stats dicts) travel as JSON bytes.  With
         `mirror_jsonl=True` (or a list of topics) every record is also appended to the files-backend
         location, so the run can be replayed with `backend: files` (`debug.streams_from`).
  redis  the reference's networked transport, wire-compatible (streams.py:106-232): XADD {index, data = pickle(record)} with
         the writer's running index, XREAD from id 0 one entry at a time, `maxlen` 1 000 000 approximate.  Needs the `redis`
         client package, imported lazily: `set_streams_backend("redis", host=, port=)` raises ImportError where it is absent
         (this image ships neither client nor server; tests/test_streams.py runs both classes against an in-process stand-in
         of the five client calls they make).  Multi-node runs use this; `shm` is one node, `files` needs a shared filesystem.
"""

from __future__ import annotations

import hashlib
import json
import logging
import os
import time
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Iterator, Literal

import numpy as np
import torch
from pydantic import BaseModel

from . import batch_codec
from .finetune.types import PipelineBatchEncoding
from .ragged import RaggedRollouts

logger = logging.getLogger(__name__)

_REREAD_DELAY = 0.1   # file backend: poll period when the tail of the file is reached
_RECHECK_DELAY = 3.0  # file backend: period of the "waiting for stream" check

_backend: str | None = None
_backend_options: dict[str, Any] = {}


def set_streams_backend(backend: Literal["files", "shm", "redis"], **kwargs: Any) -> None:
    """Select the transport once per process (reference :33-43)."""
    global _backend, _backend_options
    if _backend is not None:
        raise ValueError("Backend already set. Cannot change it.")
    if backend == "redis":
        try:
            import redis  # noqa: F401
        except ImportError as e:
            raise ImportError("streams backend 'redis' needs the `redis` client package (and a reachable server); it is not installed here. "
                              "Single-node runs can use backend 'shm' (same log semantics in shared memory) or 'files'.") from e
        unknown = set(kwargs) - {"host", "port"}
        if unknown:  # the reference's RedisConfig(host, port) rejects nothing silently either (pydantic ignores, we say so)
            raise ValueError(f"redis backend takes host and port only, got {sorted(unknown)}")
    if backend not in ("files", "shm", "redis"):
        raise ValueError(f"Invalid backend: {backend}. Only 'redis', 'files' and 'shm' are supported.")
    _backend, _backend_options = backend, dict(kwargs)


def reset_streams_backend() -> None:
    """Testing hook: forget the configured backend."""
    global _backend, _backend_options
    _backend, _backend_options = None, {}
    _checked_experiments.clear()


def unlink_shm_stream(stream: "SingleStreamSpec") -> None:
    """Remove the shared-memory log of one stream."""
    from .ring import Log

    Log.unlink_name(ring_name(stream))


def clean_shm_streams(exp_path: "str | Path") -> int:
    """Remove EVERY shared-memory log of an experiment (all topics, instances, partitions, whoever created
    them); returns the number of shm objects removed.  This is the shm counterpart of the reference launcher
    deleting `<exp_path>/streams` before a run (launch.py:463) - call it from the process that owns the run
    (launcher) at start-up, so that a run that was SIGKILLed cannot leak its records (stale `SamplesProcessed`,
    `TrainingDone`, `WeightUpdateSuccess`) into the next one, and when the run is over.  Logs are never removed
    behind the back of a live run: stream writers do NOT unlink anything when they exit."""
    prefix = _exp_prefix(exp_path)
    n = 0
    try:
        names = os.listdir("/dev/shm")
    except OSError:
        return 0
    for f in names:
        if f.startswith(prefix):
            try:
                os.unlink(os.path.join("/dev/shm", f))
                n += 1
            except OSError:
                pass
    return n


def begin_run(exp_path: "str | Path") -> int:
    """Call ONCE from the process that owns a run - the launcher, where the reference's `clean_up` removes
    `<exp_path>/streams` (launch.py:462-470) - BEFORE any stage opens a stream: every shared-memory log a previous run
    of this `exp_path` left behind is removed (writers never unlink their logs, so a finished or killed run leaves all of
    them in /dev/shm, and a new run appending to them would replay the old run's `TrainingDone`, `SamplesProcessed`,
    `WeightUpdateSuccess` from record 0), and this process removes the new run's logs again when it exits.
    Returns the number of stale shm objects removed.  A no-op for the other backends (their storage is the launcher's)."""
    if _backend not in (None, "shm"):
        return 0
    n = clean_shm_streams(exp_path)
    if n:
        logger.info(f"begin_run: removed {n} shared-memory log objects of an earlier run of {exp_path}")
    _own_experiment(exp_path)
    _checked_experiments.add(str(Path(exp_path).resolve()))
    return n


_checked_experiments: set[str] = set()


def _warn_if_logs_predate_this_process(exp_path: "str | Path") -> None:
    """First stream of an experiment opened by this process, no owner declared: if logs of that experiment already
    exist, say so once.  That is normal for a stage that joins a live run - and exactly what a forgotten
    `begin_run` / `clean_shm_streams` after a finished or killed run looks like, which is why it is worth a line."""
    key = str(Path(exp_path).resolve())
    if key in _checked_experiments or key in _owned_experiments or _backend_options.get("owner", False):
        return
    _checked_experiments.add(key)
    prefix = _exp_prefix(exp_path)
    try:
        stale = [f for f in os.listdir("/dev/shm") if f.startswith(prefix)]
    except OSError:
        return
    if stale:
        logger.warning(f"shm streams: {len(stale)} shared-memory log objects of {exp_path} exist already.  Fine if this process joins a "
                       "live run; if they are left from an EARLIER run (writers never unlink their logs), its records will be replayed "
                       "from record 0 - the process that owns a run must call pipelinerl_amd.streams.begin_run(exp_path) first "
                       "(or set_streams_backend('shm', owner=True)); see INTEGRATION.md 'launcher'.")


def raise_if_backend_not_set() -> None:
    if _backend is None:
        raise ValueError("Backend not set. Please call set_streams_backend() first.")


class SingleStreamSpec(BaseModel):
    exp_path: Path
    topic: str
    instance: int = 0
    partition: int = 0

    def __str__(self) -> str:
        return f"{self.topic}/{self.instance}/{self.partition}"


class StreamRangeSpec(BaseModel):
    exp_path: Path
    topic: str
    instance: int = 0
    partition_range: tuple[int, int]

    def __str__(self) -> str:
        return f"{self.topic}/{self.instance}/{self.partition_range[0]}-{self.partition_range[1]}"


StreamSpec = SingleStreamSpec | StreamRangeSpec


class StreamWriter(ABC):
    @abstractmethod
    def __enter__(self): ...

    @abstractmethod
    def __exit__(self, exc_type, exc_value, traceback): ...

    @abstractmethod
    def write(self, data: Any, partition: int | None = None): ...


class StreamReader(ABC):
    @abstractmethod
    def __enter__(self): ...

    @abstractmethod
    def __exit__(self, exc_type, exc_value, traceback): ...

    @abstractmethod
    def read(self) -> Iterator[Any]: ...


# ---------------------------------------------------------------------------------------------
# JSON-able view of a record (shared by both backends)
# ---------------------------------------------------------------------------------------------


def to_jsonable(data: Any) -> Any:
    """pydantic models / batches -> plain dicts; tensors and arrays -> nested lists."""
    if isinstance(data, RaggedRollouts):  # text form = the reference's list-of-dicts group record
        from .synthetic import ragged_to_entries

        return ragged_to_entries(data)
    if isinstance(data, (BaseModel, PipelineBatchEncoding)):
        data = data.model_dump()
    if isinstance(data, dict):
        return {k: to_jsonable(v) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return [to_jsonable(v) for v in data]
    if isinstance(data, torch.Tensor):
        return data.detach().cpu().tolist()
    if isinstance(data, np.ndarray):
        return data.tolist()
    if isinstance(data, np.generic):
        return data.item()
    if isinstance(data, Path):
        return str(data)
    return data


def _dumps(data: Any) -> str:
    return json.dumps(to_jsonable(data), separators=(",", ":"))


# ---------------------------------------------------------------------------------------------
# files backend
# ---------------------------------------------------------------------------------------------


def stream_dir(exp_path: Path, topic: str, instance: int, partition: int) -> Path:
    return Path(exp_path) / "streams" / topic / str(instance) / str(partition)


def stream_file(directory: Path, shard_id: int) -> Path:
    return directory / f"{shard_id}.jsonl"


class FileStreamWriter(StreamWriter):
    def __init__(self, stream: SingleStreamSpec, mode: Literal["w", "a"] = "a"):
        self.stream = stream
        self.mode = mode

    def __enter__(self):
        d = stream_dir(self.stream.exp_path, self.stream.topic, self.stream.instance, self.stream.partition)
        os.makedirs(d, exist_ok=True)
        self._file_path = stream_file(d, 0)
        self._file = open(self._file_path, self.mode)
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._file.close()

    def write(self, data, partition: int | None = None):
        if partition is not None:
            raise ValueError()
        self._file.write(_dumps(data))
        self._file.write("\n")
        self._file.flush()


class FileStreamReader(StreamReader):
    """Tails `0.jsonl` from the beginning; blocks (polling) for new complete lines forever."""

    def __init__(self, stream: SingleStreamSpec, poll_delay: float = _REREAD_DELAY):
        self.stream = stream
        self.poll_delay = poll_delay

    def __enter__(self):
        d = stream_dir(self.stream.exp_path, self.stream.topic, self.stream.instance, self.stream.partition)
        self._file_path = stream_file(d, 0)
        waited = 0.0
        while not os.path.exists(self._file_path):
            if waited % _RECHECK_DELAY < 0.05:
                logger.warning(f"Waiting for {self.stream} to be created")
            time.sleep(0.05)
            waited += 0.05
        self._file = open(self._file_path, "r")
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._file.close()

    def read(self):
        position = self._file.tell()
        retries, retry_delay = 0, 0.01
        while True:
            line = self._file.readline()
            if not line.endswith("\n"):  # tail reached (or a partially written line): rewind, wait
                self._file.seek(position)
                time.sleep(self.poll_delay)
                continue
            try:
                record = json.loads(line)
            except json.JSONDecodeError:
                # a concurrent writer can expose a torn line; reopen and retry from the same offset
                if retries >= 10:
                    logger.error(f"Error reading stream {self.stream}, giving up after {retries} retries")
                    raise
                retries += 1
                time.sleep(retry_delay)
                retry_delay *= 2
                self._file.close()
                self._file = open(self._file_path, "r")
                self._file.seek(position)
                continue
            retries, retry_delay = 0, 0.01
            position = self._file.tell()
            yield record


# ---------------------------------------------------------------------------------------------
# shm backend
# ---------------------------------------------------------------------------------------------


def _exp_prefix(exp_path: "str | Path") -> str:
    return "prl_" + hashlib.sha1(str(Path(exp_path).resolve()).encode()).hexdigest()[:12] + "_"


def ring_name(stream: SingleStreamSpec) -> str:
    """Deterministic shared-memory object name of a stream (both ends derive it from the spec):
    `prl_<experiment hash>_<stream hash>`; its segments are `<name>.<k>`.  The experiment prefix lets the
    owner of a run find all its logs (`clean_shm_streams`)."""
    key = f"{stream.topic}|{stream.instance}|{stream.partition}"
    return _exp_prefix(stream.exp_path) + hashlib.sha1(key.encode()).hexdigest()[:16]


DEFAULT_TRIM_TOPICS = ("training_data", "actor")


def _shm_options(topic: str) -> tuple[int, bool]:
    seg = int(_backend_options.get("segment_bytes", 64 << 20))
    trim_topics = _backend_options.get("trim_topics", DEFAULT_TRIM_TOPICS)
    return seg, topic in tuple(trim_topics)


_owned_experiments: set[str] = set()


def _own_experiment(exp_path: "str | Path") -> None:
    """`set_streams_backend("shm", owner=True)`: THIS process owns the runs whose streams it touches (the
    launcher, a single-process test) and removes all their logs when it exits.  Every other process leaves
    them alone: a log must outlive its writers - a reader that is still in an earlier segment, or one that
    attaches after the producer finished, has to find every record, exactly as with a file on disk or a
    Redis stream."""
    import atexit

    key = str(Path(exp_path).resolve())
    if key not in _owned_experiments:
        _owned_experiments.add(key)
        atexit.register(clean_shm_streams, key)


class ShmStreamWriter(StreamWriter):
    """Appends binary records to the stream's shared-memory log: attach if it exists (mode "a" after a
    previous writer - the reference reopens the trainer topic for every weight update), create it
    otherwise; mode "w" starts an empty log (FileStreamWriter("w") truncates the file)."""

    def __init__(self, stream: SingleStreamSpec, mode: Literal["w", "a"] = "a"):
        if mode not in ("w", "a"):
            raise ValueError(f"Invalid mode: {mode}. Only 'w' and 'a' are supported.")
        self.stream = stream
        self.mode = mode
        # `mirror_jsonl`: True or a list of topics - every record also goes to the files-backend
        # location as one JSON line, so a run can be replayed later with `backend: files`
        # (`debug.streams_from`); rollouts are mirrored as the actor's list-of-dicts record.
        mirror = _backend_options.get("mirror_jsonl", False)
        self._mirror = FileStreamWriter(stream, mode) if (mirror is True or (isinstance(mirror, (list, tuple, set)) and stream.topic in mirror)) else None

    def __enter__(self):
        from .ring import Log

        name = ring_name(self.stream)
        seg, trim = _shm_options(self.stream.topic)
        _warn_if_logs_predate_this_process(self.stream.exp_path)
        self._log = Log(name, create=True, truncate=self.mode == "w", trim=trim, segment_bytes=seg,
                        takeover_after=float(_backend_options.get("creator_timeout", 5.0)))
        if _backend_options.get("owner", False):
            _own_experiment(self.stream.exp_path)
        if self._mirror is not None:
            self._mirror.__enter__()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._log.close()
        if self._mirror is not None:
            self._mirror.__exit__(exc_type, exc_value, traceback)

    def write(self, data, partition: int | None = None):
        if partition is not None:
            raise ValueError()
        if isinstance(data, PipelineBatchEncoding):
            batch_codec.append_batch(self._log, data)  # host columns go straight into the segment (one copy)
        elif isinstance(data, RaggedRollouts):
            batch_codec.append_rollouts(self._log, data)
        else:
            self._log.append(batch_codec.encode_json(_dumps(data)))
        if self._mirror is not None:
            self._mirror.write(data.to_entries() if isinstance(data, RaggedRollouts) else data)


class ShmStreamReader(StreamReader):
    """Own cursor over the stream's log, starting at its first record; waits for the log to appear."""

    def __init__(self, stream: SingleStreamSpec):
        self.stream = stream

    def __enter__(self):
        from . import _lib
        from .ring import Log

        warned = 0.0
        _warn_if_logs_predate_this_process(self.stream.exp_path)
        while True:
            try:
                self._log = Log(ring_name(self.stream), reader=True, wait=_RECHECK_DELAY)
                if _backend_options.get("owner", False):
                    _own_experiment(self.stream.exp_path)
                return self
            except _lib.PrlError:
                if time.time() - warned > _RECHECK_DELAY:
                    logger.warning(f"Waiting for {self.stream} to be created")
                    warned = time.time()

    def __exit__(self, exc_type, exc_value, traceback):
        self._log.close()

    def read(self):
        while True:
            yield batch_codec.decode(self._log.read())


# ---------------------------------------------------------------------------------------------
# redis backend (reference streams.py:106-192; needs the `redis` client package + a server)
# ---------------------------------------------------------------------------------------------

_REDIS_RETRY_DELAY = 5.0


def _connect_to_redis():
    """Connect with unlimited retries, as the reference does (:106-117)."""
    import redis

    host, port = _backend_options.get("host", "localhost"), int(_backend_options.get("port", 6379))
    while True:
        try:
            client = redis.Redis(host=host, port=port)
            client.ping()
            return client
        except (redis.exceptions.TimeoutError, redis.ConnectionError) as e:
            logger.warning(f"Waiting for Redis server ({type(e)}). Retrying in {_REDIS_RETRY_DELAY:g} seconds.")
            time.sleep(_REDIS_RETRY_DELAY)


def _picklable(data: Any) -> Any:
    """What the reference pickles: `model_dump()` of a pydantic record (:154-155) - tensors stay tensors.  Our batch type and the
    SoA rollout record are not pydantic models; they take the same plain form (dict of tensors / the actor's list of dicts)."""
    if isinstance(data, RaggedRollouts):
        return data.to_entries()
    if isinstance(data, (BaseModel, PipelineBatchEncoding)):
        return data.model_dump()
    return data


class RedisStreamWriter(StreamWriter):
    """XADD {index, data = pickle(record)} under the stream name `topic/instance/partition`, with the running index the
    reference keeps: mode "a" continues after the last entry, mode "w" refuses a stream that already has data (:120-158)."""

    def __init__(self, stream: SingleStreamSpec, mode: Literal["w", "a"] = "a"):
        if mode not in ("w", "a"):
            raise ValueError(f"Invalid mode: {mode}. Only 'w' and 'a' are supported.")
        self.stream = stream
        self._stream_name = str(stream)
        self._redis = _connect_to_redis()
        last = self._redis.xrevrange(self._stream_name, count=1)
        if mode == "w" and last:
            raise ValueError(f"Stream {self.stream} already exists. Cannot overwrite it.")
        self._index = int(last[0][1][b"index"].decode()) + 1 if last else 0

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._redis.close()

    def write(self, data, partition: int | None = None):
        import pickle

        if partition is not None:
            raise ValueError()
        self._redis.xadd(self._stream_name, {"index": self._index, "data": pickle.dumps(_picklable(data))}, maxlen=1000000, approximate=True)
        self._index += 1


class RedisStreamReader(StreamReader):
    """XREAD from id 0, one entry per call, blocking `_REREAD_DELAY` at the tail; every entry's index must be the next one
    (a trimmed or interleaved stream raises, :163-192)."""

    def __init__(self, stream: SingleStreamSpec):
        self.stream = stream
        self._stream_name = str(stream)
        self._redis = _connect_to_redis()
        self._last_id: Any = 0
        self._index = 0

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._redis.close()

    def read(self):
        import pickle

        block = int(_REREAD_DELAY * 1000)
        while True:
            response = self._redis.xread({self._stream_name: self._last_id}, count=1, block=block)
            if not response:
                continue
            name, result = response[0]
            assert (name.decode("utf-8") if isinstance(name, bytes) else name) == self._stream_name and len(result) == 1
            entry_id, entry = result[0]
            got = int(entry[b"index"].decode("utf-8"))
            if got != self._index:
                raise ValueError(f"Index mismatch: expected {self._index}, got {got}")
            self._last_id = entry_id
            self._index += 1
            yield pickle.loads(entry[b"data"])


# ---------------------------------------------------------------------------------------------
# partitioned writer (reference :349-384)
# ---------------------------------------------------------------------------------------------


class PartitionedStreamWriter(StreamWriter):
    """One writer per partition of a `StreamRangeSpec`; `write(data, partition=k)` targets one,
    `write(data)` round-robins (reference :195-232, :349-384)."""

    def __init__(self, streams: StreamRangeSpec, mode: Literal["w", "a"], writer_cls):
        self.streams = streams
        self._next = 0
        self._writers = [
            writer_cls(SingleStreamSpec(exp_path=streams.exp_path, topic=streams.topic, instance=streams.instance, partition=i), mode=mode)
            for i in range(*streams.partition_range)
        ]

    def __enter__(self):
        for w in self._writers:
            w.__enter__()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        for w in self._writers:
            w.__exit__(exc_type, exc_value, traceback)

    def write(self, data, partition: int | None = None):
        if partition is None:
            partition = self._next
            self._next = (self._next + 1) % len(self._writers)
        elif partition < 0 or partition >= len(self._writers):
            raise ValueError(f"Invalid partition {partition}. Must be between 0 and {len(self._writers) - 1}")
        self._writers[partition].write(data)


# the reference's class names for the partitioned writers (:195-232, 349-384) and its Redis helpers (:25-30, 106-117)


class RoundRobinFileStreamWriter(PartitionedStreamWriter):
    def __init__(self, streams: StreamRangeSpec, mode: Literal["w", "a"] = "a"):
        super().__init__(streams, mode, FileStreamWriter)


class RoundRobinRedisStreamWriter(PartitionedStreamWriter):
    def __init__(self, streams: StreamRangeSpec, mode: Literal["w", "a"] = "a"):
        super().__init__(streams, mode, RedisStreamWriter)


class RedisConfig(BaseModel):
    host: str = "localhost"
    port: int = 6379


def connect_to_redis(config: "RedisConfig | None" = None):
    """Connect (unlimited retries) to `config`'s server - default: the one `set_streams_backend("redis", host=, port=)` named."""
    if config is None:
        return _connect_to_redis()
    saved = dict(_backend_options)
    _backend_options.update(host=config.host, port=config.port)
    try:
        return _connect_to_redis()
    finally:
        _backend_options.clear()
        _backend_options.update(saved)


def read_stream(stream: SingleStreamSpec) -> StreamReader:
    """Start reading the stream from the beginning."""
    raise_if_backend_not_set()
    if not isinstance(stream, SingleStreamSpec):
        raise ValueError(f"Invalid stream spec: {stream}")
    return {"files": FileStreamReader, "shm": ShmStreamReader, "redis": RedisStreamReader}[_backend](stream)


def write_to_streams(streams: StreamSpec, mode: Literal["w", "a"] = "a") -> StreamWriter:
    """Append to the end of the stream(s)."""
    raise_if_backend_not_set()
    if not isinstance(streams, (SingleStreamSpec, StreamRangeSpec)):
        raise ValueError(f"Invalid stream spec: {streams}")
    writer_cls = {"files": FileStreamWriter, "shm": ShmStreamWriter, "redis": RedisStreamWriter}[_backend]
    if isinstance(streams, SingleStreamSpec):
        return writer_cls(streams, mode)
    return PartitionedStreamWriter(streams, mode, writer_cls)
