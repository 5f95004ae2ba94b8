"""Binary wire format of stream records (shm backend).

Record = 8-byte magic/kind + payload.
  kind "PRLJSON1": UTF-8 JSON text (trainer messages, stats dicts, rollout groups).
  kind "PRLROL01": a chunk of rollouts as ragged SoA (`RaggedRollouts`: int32 tokens/labels, fp32
        completion logprobs, offsets, per-sequence scalars) - the binary form of the `actor` stream
        record, 8-16 bytes per token instead of a JSON list per field; same framing as PRLBAT01.
  kind "PRLCMP01": a packed micro-batch BEFORE expansion - the ragged columns of its sequences in packing order (int32 ids / labels,
        fp32 completion log-probs, offsets, five per-sequence scalars): 12-16 bytes per token instead of 68; the learner's
        loader runs the pack kernel on ITS device (`finetune.data.CompactBatch.to_batch`) and gets the identical batch.
        Same framing as PRLBAT01.
  kind "PRLBAT01": a PipelineBatchEncoding as SoA:
        u32 header_len | header JSON | raw buffers, each 16-byte aligned
     header = {"scalars": {model_version, sentinel, padding, is_packed},
               "tensors": [[name, dtype, shape, offset, nbytes], ...]}
     int64 / fp32 / int32 buffers are copied verbatim.  Copies per record: encode = each tensor once into the record
     buffer (a device tensor's D2H lands there directly) + the append into the shared-memory segment; decode = one
     copy out of the segment (`Log.read`) + `torch.frombuffer` VIEWS over that buffer.  No per-token Python objects on
     either side (the reference's JSONL costs ~114 bytes of text and a list->tensor conversion per token).
"""

from __future__ import annotations

import ctypes
import json
import struct
from typing import Any

import numpy as np
import torch

from .finetune.types import PipelineBatchEncoding

MAGIC_JSON = b"PRLJSON1"
MAGIC_BATCH = b"PRLBAT01"
MAGIC_ROLLOUTS = b"PRLROL01"
MAGIC_COMPACT = b"PRLCMP01"
_ALIGN = 16
_TORCH = {"int64": torch.int64, "float32": torch.float32, "int32": torch.int32, "float64": torch.float64, "uint8": torch.uint8,
          "bool": torch.bool, "bfloat16": torch.bfloat16, "float16": torch.float16}


def encode_json(text: str) -> bytes:
    return MAGIC_JSON + text.encode("utf-8")


def _frame(magic: bytes, scalars: dict, named_tensors) -> bytearray:
    """One record buffer, filled in place: the layout is fixed first, then every tensor is copied ONCE from where it
    lives (host or device - a device tensor's D2H lands directly in the record) into its 16-byte aligned slot."""
    layout = []
    offset = 0
    for name, t in named_tensors:
        t = t.detach()
        if str(t.dtype).replace("torch.", "") not in _TORCH:  # fail at the PRODUCER, not with a KeyError in the consumer process
            raise TypeError(f"stream record field {name!r} has dtype {t.dtype}; the binary record carries {sorted(_TORCH)}")
        offset += (-offset) % _ALIGN
        nbytes = t.numel() * t.element_size()
        layout.append((name, t, offset, nbytes))
        offset += nbytes
    header = json.dumps({"scalars": scalars,
                         "tensors": [[name, str(t.dtype).replace("torch.", ""), list(t.shape), off, nb] for name, t, off, nb in layout]}).encode("utf-8")
    base = 12 + len(header)
    base += (-base) % _ALIGN
    buf = bytearray(base + offset)
    buf[:8] = magic
    struct.pack_into("<I", buf, 8, len(header))
    buf[12:12 + len(header)] = header
    for _, t, off, nb in layout:
        if not nb:
            continue
        if t.device.type == "cpu" and t.is_contiguous():
            # plain memcpy: torch's copy_ hands anything above 32 K elements to its intra-op thread pool, whose wake-up
            # costs more than the copy (measured 70 ms per 256 KB column under a CPU quota)
            np.frombuffer(buf, dtype=np.uint8, count=nb, offset=base + off)[:] = t.reshape(-1).view(torch.uint8).numpy()  # reshape first: 0-dim tensors have no byte view
        else:
            torch.frombuffer(buf, dtype=t.dtype, count=t.numel(), offset=base + off).view(t.shape).copy_(t)
    return buf


def append_framed(log, magic: bytes, scalars: dict, named_tensors) -> int:
    """The record `_frame` would build, written by `log.appendv` STRAIGHT into the shared-memory segment: the 12-byte
    prefix + header JSON from a small host buffer, every contiguous CPU tensor from where it lies (one copy per byte,
    source -> segment; `_frame` + `append` make it source -> record buffer -> segment, plus the buffer's zero fill).
    Tensors that are not contiguous CPU tensors (device tensors, strided views) go through `_frame`.  Returns the
    record size."""
    tensors = [(name, t.detach()) for name, t in named_tensors]
    if any(t.device.type != "cpu" or not t.is_contiguous() for _, t in tensors):
        buf = _frame(magic, scalars, tensors)
        log.append(buf)
        return len(buf)
    layout, offset = [], 0
    for name, t in tensors:
        if str(t.dtype).replace("torch.", "") not in _TORCH:
            raise TypeError(f"stream record field {name!r} has dtype {t.dtype}; the binary record carries {sorted(_TORCH)}")
        offset += (-offset) % _ALIGN
        nbytes = t.numel() * t.element_size()
        layout.append((name, t, offset, nbytes))
        offset += nbytes
    header = json.dumps({"scalars": scalars,
                         "tensors": [[name, str(t.dtype).replace("torch.", ""), list(t.shape), off, nb] for name, t, off, nb in layout]}).encode("utf-8")
    head = bytearray(12 + len(header))
    head[:8] = magic
    struct.pack_into("<I", head, 8, len(header))
    head[12:] = header
    base = len(head) + (-len(head)) % _ALIGN
    head_c = (ctypes.c_char * len(head)).from_buffer(head)
    pieces = [(ctypes.addressof(head_c), 0, len(head))]
    pieces += [(t.data_ptr(), base + off, nb) for _, t, off, nb in layout if nb]
    log.appendv(pieces, base + offset, keep_alive=(head, tensors))
    return base + offset


def describe_batch(batch: PipelineBatchEncoding, block_ptr: int, block_nbytes: int, inline: bytearray):
    """The record `append_batch` would write, as a recipe for the native publisher (csrc/prl_publish.cpp): returns
    (record size, [(kind, src, offset in the record, nbytes)]) with kind 0 = a range of the packed block at `block_ptr` (device
    memory now, page-locked host memory by the time the publisher gathers the record), kind 1 = a range of `inline`, to which
    the record's header and its host-resident tensors (seq_boundaries, a sentinel batch) are appended here.  Same framing and
    byte layout as `_frame` / `append_framed`: a reader cannot tell which path wrote a record."""
    scalars = {"model_version": batch.model_version, "sentinel": batch.sentinel, "padding": batch.padding, "is_packed": batch.is_packed}
    layout, offset = [], 0
    for name, t in batch.tensors():
        t = t.detach()
        dt = str(t.dtype).replace("torch.", "")
        if dt not in _TORCH:
            raise TypeError(f"stream record field {name!r} has dtype {t.dtype}; the binary record carries {sorted(_TORCH)}")
        if not t.is_contiguous():
            raise ValueError(f"stream record field {name!r} is not contiguous")
        offset += (-offset) % _ALIGN
        nbytes = t.numel() * t.element_size()
        layout.append((name, dt, t, offset, nbytes))
        offset += nbytes
    header = json.dumps({"scalars": scalars, "tensors": [[name, dt, list(t.shape), off, nb] for name, dt, t, off, nb in layout]}).encode("utf-8")
    pieces = []
    at = len(inline)
    inline += MAGIC_BATCH + struct.pack("<I", len(header)) + header
    head_len = 12 + len(header)
    pieces.append((1, at, 0, head_len))
    base = head_len + (-head_len) % _ALIGN
    for name, _, t, off, nb in layout:
        if not nb:
            continue
        if t.device.type == "cpu":
            at = len(inline)
            inline += t.reshape(-1).view(torch.uint8).numpy().tobytes()
            pieces.append((1, at, base + off, nb))
        else:
            src = t.data_ptr() - block_ptr
            if src < 0 or src + nb > block_nbytes:
                raise ValueError(f"stream record field {name!r} does not live in the packed block")
            pieces.append((0, src, base + off, nb))
    return base + offset, pieces


def append_batch(log, batch: PipelineBatchEncoding) -> int:
    scalars = {"model_version": batch.model_version, "sentinel": batch.sentinel, "padding": batch.padding, "is_packed": batch.is_packed}
    return append_framed(log, MAGIC_BATCH, scalars, list(batch.tensors()))


def append_rollouts(log, rollouts) -> int:
    tensors = [(name, getattr(rollouts, name)) for name in _ROLLOUT_FIELDS if getattr(rollouts, name) is not None]
    return append_framed(log, MAGIC_ROLLOUTS, {"group_ids": list(rollouts.group_ids)}, tensors)


_ROLLOUT_FIELDS = ("tokens", "labels", "logprobs", "ref_logprobs", "seq_off", "lp_off", "reward", "group_index", "step_index",
                   "rollout_index", "model_version", "finished", "finish_code")


def encode_rollouts(rollouts) -> bytearray:
    """RaggedRollouts -> one binary record."""
    tensors = [(name, getattr(rollouts, name)) for name in _ROLLOUT_FIELDS if getattr(rollouts, name) is not None]
    return _frame(MAGIC_ROLLOUTS, {"group_ids": list(rollouts.group_ids)}, tensors)


_COMPACT_FIELDS = ("tokens", "labels", "logprobs", "ref_logprobs", "seq_off", "lp_off", "seq_scalars", "ref_column")


def compact_layout(n: int, nc: int, m: int, has_ref: bool, model_version: int, padding: int, eos_token_id: int,
                   slice_index: int = 0, num_slices: int = 1, ref_column_tokens: int = 0):
    """(header bytes incl. magic and length, offset of the first tensor, {name: (offset from there, nbytes)}, record size) of the
    compact record of a micro-batch of `n` tokens / `nc` completion tokens / `m` sequences.  One definition for the generic
    encoder (`encode_compact`) and for the preprocessor's gather recipe (the native publisher copies every column's per-sequence
    slices straight from the decoded `actor` records).  The header text is formatted by hand - the preprocessor builds one per
    micro-batch - and is the text `json.dumps` produces for the same structure (tests/test_compact_wire_host.py).
    `num_slices` > 1 (sequence parallelism): the record names the token slice its reader keeps after expansion (`slice`, `slices`
    scalars; absent otherwise, so the records of the common case are unchanged).  `ref_column_tokens` > 0 (a reference policy in the
    preprocessor, KL on): one more fp32 column of that many entries - the expanded batch's `ref_logprobs`, token for token."""
    shapes = [("tokens", "int32", f"[{n}]", 4 * n), ("labels", "int32", f"[{n}]", 4 * n), ("logprobs", "float32", f"[{nc}]", 4 * nc)]
    if has_ref:
        shapes.append(("ref_logprobs", "float32", f"[{nc}]", 4 * nc))
    shapes += [("seq_off", "int64", f"[{m + 1}]", 8 * (m + 1)), ("lp_off", "int64", f"[{m + 1}]", 8 * (m + 1)), ("seq_scalars", "float32", f"[5, {m}]", 20 * m)]
    if ref_column_tokens:
        shapes.append(("ref_column", "float32", f"[{ref_column_tokens}]", 4 * ref_column_tokens))
    offset, tensors, where = 0, [], {}
    for name, dt, shape, nb in shapes:
        offset += (-offset) % _ALIGN
        tensors.append(f'["{name}", "{dt}", {shape}, {offset}, {nb}]')
        where[name] = (offset, nb)
        offset += nb
    sliced = ', "slice": %d, "slices": %d' % (int(slice_index), int(num_slices)) if num_slices > 1 else ""
    header = ('{"scalars": {"model_version": %d, "padding": %d, "eos_token_id": %d%s}, "tensors": [%s]}'
              % (int(model_version), int(padding), int(eos_token_id), sliced, ", ".join(tensors))).encode("utf-8")
    head = MAGIC_COMPACT + struct.pack("<I", len(header)) + header
    base = len(head) + (-len(head)) % _ALIGN
    return head, base, where, base + offset


def encode_compact(cb) -> bytearray:
    """`finetune.data.CompactBatch` -> one record (the generic path: every column copied once into a record buffer)."""
    t = torch.from_numpy
    tensors = [(k, t(np.ascontiguousarray(getattr(cb, k)))) for k in _COMPACT_FIELDS if getattr(cb, k) is not None]
    scalars = {"model_version": int(cb.model_version), "padding": int(cb.padding), "eos_token_id": int(cb.eos_token_id)}
    if cb.num_slices > 1:
        scalars.update(slice=int(cb.slice_index), slices=int(cb.num_slices))
    return _frame(MAGIC_COMPACT, scalars, tensors)


def encode_batch(batch: PipelineBatchEncoding) -> bytearray:
    scalars = {"model_version": batch.model_version, "sentinel": batch.sentinel, "padding": batch.padding, "is_packed": batch.is_packed}
    return _frame(MAGIC_BATCH, scalars, list(batch.tensors()))


def decode(record: "bytes | bytearray") -> Any:
    """record -> dict (JSON records; batch records decode to the kwargs of PipelineBatchEncoding, exactly what the
    file backend's `json.loads` line gives `PipelineBatchEncoding(**d)`, but with tensors instead of nested lists).

    The tensors are VIEWS into `record` (`torch.frombuffer`, no per-tensor copy): pass the `bytearray` that
    `ring.Log.read` returns - the one copy out of the shared-memory segment, whose mapping the reader releases when it
    moves on - and keep nothing else alive.  An immutable `bytes` record is copied once first (torch needs a writable
    buffer)."""
    magic = bytes(record[:8])
    if magic == MAGIC_JSON:
        return json.loads(bytes(record[8:]).decode("utf-8"))
    if magic not in (MAGIC_BATCH, MAGIC_ROLLOUTS, MAGIC_COMPACT):
        raise ValueError(f"unknown record kind {magic!r}")
    if not isinstance(record, bytearray):
        record = bytearray(record)
    (hlen,) = struct.unpack_from("<I", record, 8)
    header = json.loads(bytes(record[12: 12 + hlen]).decode("utf-8"))
    base = 12 + hlen
    base += (-base) % _ALIGN
    out: dict[str, Any] = dict(header["scalars"])
    for name, dtype, shape, off, nbytes in header["tensors"]:
        if nbytes == 0:
            out[name] = torch.empty(shape, dtype=_TORCH[dtype])
            continue
        dt = _TORCH[dtype]
        out[name] = torch.frombuffer(record, dtype=dt, count=nbytes // dt.itemsize, offset=base + off).view(shape)
    if magic == MAGIC_COMPACT:
        from .finetune.data import CompactBatch

        return CompactBatch(**{k: (out[k].numpy() if k in out else None) for k in _COMPACT_FIELDS},
                            model_version=out["model_version"], padding=out["padding"], eos_token_id=out["eos_token_id"],
                            slice_index=out.get("slice", 0), num_slices=out.get("slices", 1))
    if magic == MAGIC_ROLLOUTS:
        from .ragged import RaggedRollouts

        np_of = lambda k: out[k].numpy() if k in out else None  # noqa: E731
        return RaggedRollouts.from_numpy(*[np_of(k) for k in _ROLLOUT_FIELDS], group_ids=out.get("group_ids"))
    return out
