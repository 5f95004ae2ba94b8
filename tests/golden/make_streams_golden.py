"""Golden stream files written by EXECUTING the reference's own `files` backend.

`pipelinerl/streams.py` imports `orjson` and `redis`, neither of which is installed here.  This
script registers two stand-in modules before importing it:

  * `orjson`: `dumps(obj, option=OPT_SERIALIZE_NUMPY)` implemented with the stdlib encoder (compact
    separators, numpy arrays/scalars as lists/numbers).  The directory layout, the file naming, the
    record structure (what `FileStreamWriter.write` does to pydantic models and tensors), the
    round-robin / explicit partition rule and the newline-per-record framing all come from the
    reference's code; only number formatting inside a line is the stand-in's (so tests compare parsed
    records, not bytes).
  * `typing.Self` is back-filled from typing_extensions (this image runs python 3.10).
  * `redis`: an empty module with the `exceptions` names the import needs (the redis backend is not run).

Output: tests/golden/streams_files.json = {"files": {relative path: text}, "records": {...}} plus the
list of records the reference's own reader yields from each file.

    python tests/golden/make_streams_golden.py
"""

from __future__ import annotations

import json
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent


def install_stubs() -> None:
    orjson = types.ModuleType("orjson")
    orjson.OPT_SERIALIZE_NUMPY = 16

    def default(o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, np.generic):
            return o.item()
        if isinstance(o, Path):
            return str(o)
        raise TypeError(f"not serializable: {type(o)}")

    orjson.dumps = lambda obj, option=0: json.dumps(obj, separators=(",", ":"), default=default).encode("utf-8")
    sys.modules["orjson"] = orjson
    redis = types.ModuleType("redis")
    exc = types.ModuleType("redis.exceptions")
    exc.TimeoutError = type("TimeoutError", (Exception,), {})
    redis.ConnectionError = type("ConnectionError", (Exception,), {})
    redis.exceptions = exc
    redis.Redis = object
    sys.modules["redis"] = redis
    sys.modules["redis.exceptions"] = exc
    import typing

    if not hasattr(typing, "Self"):  # the reference targets python >= 3.11
        import typing_extensions

        typing.Self = typing_extensions.Self


def scenario(streams, exp: Path) -> dict:
    """The writes every stage of the pipeline performs, in miniature."""
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

    # (1) a single stream of plain dicts (actor -> preprocessor: lists of training texts)
    with streams.write_to_streams(streams.SingleStreamSpec(exp_path=exp, topic="actor")) as w:
        w.write([{"input_ids": [1, 2, 3], "labels": [-100, 2, 3], "reward": 1.0, "logprobs": [-0.5, -0.25], "metadata": {"group_id": "g0"}}])
        w.write([{"input_ids": [4], "labels": [4], "reward": 0.0, "logprobs": [-1.5], "metadata": {"group_id": "g1", "nested": [1, {"a": None}]}}])
    # (2) partition range: round robin and explicit partitions (preprocessor -> trainers)
    rng = streams.StreamRangeSpec(exp_path=exp, topic="training_data", partition_range=(0, 3))
    with streams.write_to_streams(rng) as w:
        for i in range(5):
            w.write({"i": i})
        w.write({"i": "explicit"}, partition=2)
        w.write(WithTensor(input_ids=torch.tensor([[5, 6, 7]]), rewards=torch.tensor([[0.5, 0.0, 1.0]])), partition=1)
    # (3) pydantic control messages, non-default instance, append mode across two opens
    spec = streams.SingleStreamSpec(exp_path=exp, topic="stats", instance=2, partition=1)
    with streams.write_to_streams(spec) as w:
        w.write(Msg(samples_processed=8))
    with streams.write_to_streams(spec) as w:
        w.write(Msg(samples_processed=16))
    # (4) mode "w" truncates
    spec_w = streams.SingleStreamSpec(exp_path=exp, topic="weight_update_request")
    with streams.write_to_streams(spec_w) as w:
        w.write({"version": 1})
    with streams.write_to_streams(spec_w, mode="w") as w:
        w.write({"version": 2, "np": np.arange(3)})
    return {"str_single": str(streams.SingleStreamSpec(exp_path=exp, topic="stats", instance=2, partition=1)), "str_range": str(rng)}


def main() -> None:
    install_stubs()
    sys.path.insert(0, "/root/reference")
    from pipelinerl import streams

    streams.set_streams_backend("files")
    with tempfile.TemporaryDirectory() as td:
        exp = Path(td)
        meta = scenario(streams, exp)
        files = {str(p.relative_to(exp)): p.read_text() for p in sorted(exp.rglob("*")) if p.is_file()}
        # what the reference's own reader yields (complete lines only)
        records = {}
        for rel in files:
            parts = Path(rel).parts  # streams/<topic>/<instance>/<partition>/0.jsonl
            spec = streams.SingleStreamSpec(exp_path=exp, topic=parts[1], instance=int(parts[2]), partition=int(parts[3]))
            n = files[rel].count("\n")
            got = []
            with streams.read_stream(spec) as r:
                it = r.read()
                for _ in range(n):
                    got.append(next(it))
            records[rel] = got
    out = {"files": files, "records": records, **meta}
    (HERE / "streams_files.json").write_text(json.dumps(out, indent=1, sort_keys=True))
    print(f"wrote streams_files.json: {len(files)} files")
    for k, v in files.items():
        print(k, repr(v[:100]))


if __name__ == "__main__":
    main()
