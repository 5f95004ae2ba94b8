"""Binary wire format of stream records (shm backend).

Record = 8-byte magic/kind + payload.
  kind "PRLJSON1": UTF-8 JSON text (trainer messages, stats dicts, rollout groups).
  kind "PRLROL01": a chunk of rollouts as ragged SoA (`RaggedRollouts`: int32 tokens/labels, fp32
        completion logprobs, offsets, per-sequence scalars) - the binary form of the `actor` stream
        record, 8-16 bytes per token instead of a JSON list per field; same framing as PRLBAT01.
  kind "PRLBAT01": a PipelineBatchEncoding as SoA:
        u32 header_len | header JSON | raw buffers, each 16-byte aligned
     header = {"scalars": {model_version, sentinel, padding, is_packed},
               "tensors": [[name, dtype, shape, offset, nbytes], ...]}
     int64 / fp32 / int32 buffers are copied verbatim, so decoding is `torch.frombuffer` views over
     one bytes object: no per-token Python objects on either side (the reference's JSONL costs
     ~114 bytes of text and a list->tensor conversion per token).
"""

from __future__ import annotations

import json
import struct
from typing import Any

import numpy as np
import torch

from .finetune.types import PipelineBatchEncoding

MAGIC_JSON = b"PRLJSON1"
MAGIC_BATCH = b"PRLBAT01"
MAGIC_ROLLOUTS = b"PRLROL01"
_ALIGN = 16
_TORCH = {"int64": torch.int64, "float32": torch.float32, "int32": torch.int32, "float64": torch.float64, "uint8": torch.uint8}


def encode_json(text: str) -> bytes:
    return MAGIC_JSON + text.encode("utf-8")


def _frame(magic: bytes, scalars: dict, named_arrays) -> bytes:
    tensors = []
    blobs = []
    offset = 0
    for name, a in named_arrays:
        a = np.ascontiguousarray(a)
        raw = a.tobytes()
        pad = (-offset) % _ALIGN
        if pad:
            blobs.append(b"\0" * pad)
            offset += pad
        tensors.append([name, str(a.dtype), list(a.shape), offset, len(raw)])
        blobs.append(raw)
        offset += len(raw)
    header = json.dumps({"scalars": scalars, "tensors": tensors}).encode("utf-8")
    head = magic + struct.pack("<I", len(header)) + header
    head += b"\0" * ((-len(head)) % _ALIGN)
    return head + b"".join(blobs)


_ROLLOUT_FIELDS = ("tokens", "labels", "logprobs", "ref_logprobs", "seq_off", "lp_off", "reward", "group_index", "step_index",
                   "rollout_index", "model_version", "finished", "finish_code")


def encode_rollouts(rollouts) -> bytes:
    """RaggedRollouts -> one binary record."""
    arrays = []
    for name in _ROLLOUT_FIELDS:
        t = getattr(rollouts, name)
        if t is not None:
            arrays.append((name, t.detach().cpu().numpy()))
    return _frame(MAGIC_ROLLOUTS, {"group_ids": list(rollouts.group_ids)}, arrays)


def encode_batch(batch: PipelineBatchEncoding) -> bytes:
    tensors = []
    blobs = []
    offset = 0
    for name, t in batch.tensors():
        a = t.detach().cpu().contiguous().numpy()
        raw = a.tobytes()
        pad = (-offset) % _ALIGN
        if pad:
            blobs.append(b"\0" * pad)
            offset += pad
        tensors.append([name, str(a.dtype), list(a.shape), offset, len(raw)])
        blobs.append(raw)
        offset += len(raw)
    header = json.dumps({
        "scalars": {"model_version": batch.model_version, "sentinel": batch.sentinel, "padding": batch.padding, "is_packed": batch.is_packed},
        "tensors": tensors,
    }).encode("utf-8")
    head = MAGIC_BATCH + struct.pack("<I", len(header)) + header
    head += b"\0" * ((-len(head)) % _ALIGN)
    return head + b"".join(blobs)


def decode(record: bytes) -> Any:
    """bytes -> dict (JSON records; batch records decode to the kwargs of PipelineBatchEncoding,
    exactly what the file backend's `json.loads` line gives `PipelineBatchEncoding(**d)`, but with
    tensors instead of nested lists)."""
    magic = record[:8]
    if magic == MAGIC_JSON:
        return json.loads(record[8:].decode("utf-8"))
    if magic not in (MAGIC_BATCH, MAGIC_ROLLOUTS):
        raise ValueError(f"unknown record kind {magic!r}")
    (hlen,) = struct.unpack_from("<I", record, 8)
    header = json.loads(record[12 : 12 + hlen].decode("utf-8"))
    base = 12 + hlen
    base += (-base) % _ALIGN
    out: dict[str, Any] = dict(header["scalars"])
    buf = memoryview(record)
    for name, dtype, shape, off, nbytes in header["tensors"]:
        if nbytes == 0:
            out[name] = torch.empty(shape, dtype=_TORCH[dtype])
            continue
        a = np.frombuffer(buf, dtype=np.dtype(dtype), count=nbytes // np.dtype(dtype).itemsize, offset=base + off)
        out[name] = torch.from_numpy(a.reshape(shape).copy())
    if magic == MAGIC_ROLLOUTS:
        from .ragged import RaggedRollouts

        np_of = lambda k: out[k].numpy() if k in out else None  # noqa: E731
        return RaggedRollouts.from_numpy(*[np_of(k) for k in _ROLLOUT_FIELDS], group_ids=out.get("group_ids"))
    return out
