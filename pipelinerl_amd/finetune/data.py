"""Sample preparation and pad / pack collation (drop-in for reference
pipelinerl/finetune/data.py:111-283) on top of the K6/K7 kernels.

Two levels:

* `pack_prepared` / `pad_prepared`: the MI355X path.  Inputs are `PreparedRollouts` (ragged SoA
  buffers + per-sequence scalars, already in HBM); one launch writes a whole optimizer step's
  micro-batches back to back and the returned `PipelineBatchEncoding`s are views.
* `collate_packed` / `collate` / `preprocess_fn`: the reference's list-of-dicts signatures, for
  callers that still hold python lists.  They flatten the lists to ragged arrays, upload, and
  run the same kernels (per-token columns are honoured through `per_token_columns`).
"""

from __future__ import annotations

import contextlib
import itertools

import logging
from dataclasses import dataclass
from typing import Any, Iterable, Sequence

import numpy as np
import torch

from .. import _lib
from ..ragged import RaggedRollouts
from .rl import RL_DATA_COLUMNS, PreparedRollouts, prepare_rl_fields
from .types import PipelineBatchEncoding
from .utils import create_sentinel_example

logger = logging.getLogger(__name__)

MASKED_TOKEN_ID = -100  # ignore_index of the LM loss

# bit positions of `per_token_columns` (include/prl.h)
_PER_TOKEN_BITS = {"rewards": 1, "advantages": 2, "group_tokens": 4, "num_labels": 8, "overflow": 16}


# ---------------------------------------------------------------------------------------------
# device level
# ---------------------------------------------------------------------------------------------


_F32_COLUMNS = ("rewards", "advantages", "ref_logprobs", "old_logprobs", "group_tokens", "num_labels", "overflow")
# tensor order of a packed batch's stream record = the order `PipelineBatchEncoding.tensors()` yields them (types._FIELD_ORDER)
_RECORD_ORDER = ("input_ids", "attention_mask", "labels", "position_ids", "segment_ids", "rewards", "advantages", "ref_logprobs", "old_logprobs",
                 "group_tokens", "num_labels", "overflow", "seq_boundaries")


_I64_COLUMNS = ("input_ids", "labels", "attention_mask", "position_ids", "segment_ids")  # block order of a packed launch (_column_views)
_N_I64 = len(_I64_COLUMNS)
_TEMPLATES: dict = {}


def _record_template(n: int, n_bounds: int, model_version: int, padding: int):
    """Header bytes and layout of the stream record of a packed micro-batch of `n` tokens / `n_bounds` boundaries: everything
    about a record that does not depend on WHERE in the block its columns lie.  Cached: a run sees few distinct shapes per
    model version, and building the header (a JSON dump of 13 tensor entries) is most of what a record costs the host."""
    import json
    import struct

    key = (n, n_bounds, model_version, padding)
    got = _TEMPLATES.get(key)
    if got is not None:
        return got
    layout, offset, tensors = [], 0, []
    for name in _RECORD_ORDER:
        offset += (-offset) % 16
        if name == "seq_boundaries":
            nb = 4 * n_bounds
            tensors.append([name, "int32", [n_bounds], offset, nb])
            layout.append((-1, 4, offset, nb))
        elif name in _I64_COLUMNS:
            nb = 8 * n
            tensors.append([name, "int64", [1, n], offset, nb])
            layout.append((_I64_COLUMNS.index(name), 8, offset, nb))
        else:
            nb = 4 * n
            tensors.append([name, "float32", [1, n], offset, nb])
            layout.append((_F32_COLUMNS.index(name), 4, offset, nb))
        offset += nb
    scalars = {"model_version": model_version, "sentinel": False, "padding": padding, "is_packed": True}
    header = json.dumps({"scalars": scalars, "tensors": tensors}).encode("utf-8")
    head = b"PRLBAT01" + struct.pack("<I", len(header)) + header
    base = len(head) + (-len(head)) % 16
    if len(_TEMPLATES) > 4096:
        _TEMPLATES.clear()
    got = _TEMPLATES[key] = (head, base, tuple(layout), base + offset)
    return got


def _column_stride(total: int) -> int:
    """Elements between two columns of a block: `total` rounded up to 64, so that every column starts on a 256-byte
    boundary whatever the token count (the pack kernels store 16 bytes per lane)."""
    return (total + 63) // 64 * 64


def _column_views(block: torch.Tensor, total: int, packed: bool) -> dict[str, torch.Tensor]:
    """The batch columns as views into one uint8 block: the int64 columns first, then the fp32 ones, each `total` long."""
    names_i64 = ["input_ids", "labels", "attention_mask"] + (["position_ids", "segment_ids"] if packed else [])
    n64, stride = len(names_i64), _column_stride(total)
    i64 = block[: n64 * stride * 8].view(torch.int64).view(n64, stride)
    f32 = block[n64 * stride * 8:].view(torch.float32).view(len(_F32_COLUMNS), stride)
    out = {k: i64[i, :total] for i, k in enumerate(names_i64)}
    out.update({k: f32[i, :total] for i, k in enumerate(_F32_COLUMNS)})
    return out


def _alloc_outputs(total: int, dev: torch.device, packed: bool) -> dict[str, torch.Tensor]:
    """ONE allocation for the 12 (packed) / 10 (padded) output columns - 68 / 52 bytes per token - so that a whole launch's
    output can leave the device in one copy (`PackedStep.to_host`); `out["__block__"]` is that allocation."""
    n64 = 5 if packed else 3
    block = torch.empty(_column_stride(total) * (n64 * 8 + len(_F32_COLUMNS) * 4), dtype=torch.uint8, device=dev)
    out = _column_views(block, total, packed)
    out["__block__"] = block
    return out


class PackedStep(Sequence):
    """Result of one K6 launch: flat per-token buffers holding consecutive micro-batches.  Behaves
    like a list of `PipelineBatchEncoding`; the per-micro-batch objects (12 tensor views each) are
    only materialised when indexed, so a 4096-micro-batch step costs no host time up front."""

    def __init__(self, flat: dict[str, torch.Tensor], pk_dst: np.ndarray, mb_off: np.ndarray,
                 model_versions: np.ndarray, pads: np.ndarray | None):
        flat = dict(flat)
        self.block = flat.pop("__block__", None)  # the one allocation behind every column (None: separate tensors)
        self.flat = flat
        self.pk_dst = pk_dst          # int64 [m + 1] token offset of every packed sequence
        self.mb_off = mb_off          # int64 [n_mb + 1] index into pk_dst of every micro-batch
        self.model_versions = model_versions  # int64 [n_mb]
        self.pads = pads
        self.token_off = pk_dst[mb_off]  # int64 [n_mb + 1] token offset of every micro-batch
        self._cache: dict[int, PipelineBatchEncoding] = {}

    def __len__(self) -> int:
        return len(self.mb_off) - 1

    @property
    def total_tokens(self) -> int:
        return int(self.pk_dst[-1])

    def __getitem__(self, j):  # type: ignore[override]
        if isinstance(j, slice):
            return [self[i] for i in range(*j.indices(len(self)))]
        if j < 0:
            j += len(self)
        got = self._cache.get(j)
        if got is None:
            a, b = int(self.mb_off[j]), int(self.mb_off[j + 1])
            t0, t1 = int(self.pk_dst[a]), int(self.pk_dst[b])
            bounds = (self.pk_dst[a : b + 1] - self.pk_dst[a]).astype(np.int32)
            got = PipelineBatchEncoding(
                **{k: v[t0:t1].unsqueeze(0) for k, v in self.flat.items()},
                model_version=int(self.model_versions[j]),
                is_packed=True,
                seq_boundaries=torch.from_numpy(bounds),
                padding=int(self.pads[j]) if self.pads is not None else 0,
            )
            self._cache[j] = got
        return got

    def to_host(self, stager) -> "PackedStep":
        """The same micro-batches over HOST memory: the whole block leaves the device in ONE copy into a page-locked
        buffer of `stager` (a `staging.PinnedStager`); the columns are views of that buffer - consume them (encode them
        into stream records) before the stager's ring comes round."""
        if self.block is None:
            raise RuntimeError("this PackedStep was not allocated as one block")
        host = stager.download(self.block)
        flat = _column_views(host, self.total_tokens, packed=True)
        return PackedStep(flat, self.pk_dst, self.mb_off, self.model_versions, self.pads)

    def describe_record(self, j: int, inline: bytearray) -> tuple[int, list[tuple[int, int, int, int]]]:
        """`batch_codec.describe_batch(self[j], ...)` by arithmetic on the block's geometry alone - no tensor views, no per-column
        torch calls (12 slices + 12 pointer queries per micro-batch are ~70 us of host time, more than the publisher thread
        needs to move the record): the recipe of micro-batch j's stream record for the native publisher.  Returns (record size,
        [(kind, src, offset in the record, nbytes)]), kind 0 = a byte range of `self.block`, kind 1 = a range of `inline`
        (header and seq_boundaries are appended there).  Same bytes as the generic path (tests/test_streams.py)."""
        if self.block is None:
            raise RuntimeError("this PackedStep was not allocated as one block")
        a, b = int(self.mb_off[j]), int(self.mb_off[j + 1])
        t0, n = int(self.pk_dst[a]), int(self.pk_dst[b] - self.pk_dst[a])
        bounds = (self.pk_dst[a: b + 1] - self.pk_dst[a]).astype(np.int32)
        head, base, layout, total = _record_template(n, len(bounds), int(self.model_versions[j]), int(self.pads[j]) if self.pads is not None else 0)
        stride = _column_stride(self.total_tokens)
        at = len(inline)
        inline += head
        pieces = [(1, at, 0, len(head))]
        for col, size, off, nb in layout:
            if not nb:
                continue
            if col < 0:  # seq_boundaries: host data
                at = len(inline)
                inline += bounds.tobytes()
                pieces.append((1, at, base + off, nb))
            elif size == 8:
                pieces.append((0, (col * stride + t0) * 8, base + off, nb))
            else:
                pieces.append((0, _N_I64 * stride * 8 + (col * stride + t0) * 4, base + off, nb))
        return total, pieces

    def step_batch(self) -> PipelineBatchEncoding:
        """The whole launch as ONE [1, T_step] batch over the same buffers (no copy)."""
        return PipelineBatchEncoding(
            **{k: v.unsqueeze(0) for k, v in self.flat.items()},
            model_version=int(self.model_versions.min()) if len(self.model_versions) else 0, is_packed=True,
        )


def plan_packing(lens: np.ndarray, micro_batches: Sequence[Sequence[int]], sentinel_pad: Sequence[int] | None):
    """Host-side O(#sequences) plan of one K6 launch, vectorised: (pk_src, pk_seg, pk_dst, mb_off)."""
    n_mb = len(micro_batches)
    # (C-level iteration: a generator expression over 4096 one-element lists cost more than the kernel's planning is worth)
    counts = np.fromiter(map(len, micro_batches), dtype=np.int64, count=n_mb)
    src = np.fromiter(itertools.chain.from_iterable(micro_batches), dtype=np.int64, count=int(counts.sum()))
    pads = None if sentinel_pad is None else np.asarray(sentinel_pad, dtype=np.int64)
    if pads is None or not pads.any():
        mb_off = np.zeros(n_mb + 1, dtype=np.int64)
        np.cumsum(counts, out=mb_off[1:])
        seg = np.arange(len(src), dtype=np.int64) - np.repeat(mb_off[:-1], counts)
        pk_len = lens[src] if len(src) else np.zeros(0, dtype=np.int64)
        pk_src = src
    else:
        extra = (pads > 0).astype(np.int64)
        counts2 = counts + extra
        mb_off = np.zeros(n_mb + 1, dtype=np.int64)
        np.cumsum(counts2, out=mb_off[1:])
        m = int(mb_off[-1])
        pk_src = np.full(m, -1, dtype=np.int64)
        pk_len = np.zeros(m, dtype=np.int64)
        seg = np.arange(m, dtype=np.int64) - np.repeat(mb_off[:-1], counts2)
        real = seg < np.repeat(counts, counts2)  # the filler, when present, is the last slot
        pk_src[real] = src
        pk_len[real] = lens[src]
        pk_len[~real] = pads[extra > 0]
    pk_dst = np.zeros(len(pk_src) + 1, dtype=np.int64)
    np.cumsum(pk_len, out=pk_dst[1:])
    return pk_src.astype(np.int32), seg.astype(np.int32), pk_dst, mb_off


def pack_prepared(
    prep: PreparedRollouts,
    micro_batches: Sequence[Sequence[int]],
    eos_token_id: int,
    sentinel_pad: Sequence[int] | None = None,
    per_token_columns: int = 0,
    timer: Any = None,
    stager: Any = None,
) -> PackedStep:
    """Pack `micro_batches[j]` (lists of sequence indices into `prep`) into packed batches with a
    single K6 launch.  `sentinel_pad[j]` > 0 appends that many filler tokens to micro-batch j
    (sequence-parallel padding, reference data.py:222-230).  Returns a list-like `PackedStep`;
    every micro-batch is a view into shared flat buffers."""
    lib = _lib.load()
    r = prep.rollouts
    dev = r.device
    lens = r.seq_lengths()
    pk_src, pk_seg, pk_dst, mb_off = plan_packing(lens, micro_batches, sentinel_pad)
    m = len(pk_src)
    total = int(pk_dst[-1])
    out = _alloc_outputs(total, dev, packed=True)
    if stager is not None:  # the three plan arrays in one page-locked copy
        d_src, d_dst, d_seg = stager.upload([pk_src, pk_dst, pk_seg])
    else:
        from .rl import upload_packed

        d_src, d_dst, d_seg = upload_packed([pk_src, pk_dst, pk_seg], dev)  # one copy
    if m and total:
        # `timer` (bench.py's EventTimer): HIP events around the kernel alone, so that the host planning
        # above (O(#sequences) numpy + three small uploads) is not charged to the kernel's bandwidth
        with torch.cuda.device(dev), (timer.time("pack_collate_kernel") if timer is not None else contextlib.nullcontext()):
            _lib.check(
                lib.prl_pack_collate(
                    m, total, _lib.ptr(d_src), _lib.ptr(d_dst), _lib.ptr(d_seg), _lib.ptr(r.tokens),
                    _lib.ptr(r.labels), _lib.ptr(r.logprobs), _lib.ptr(r.ref_logprobs), _lib.ptr(r.seq_off),
                    _lib.ptr(r.lp_off), _lib.ptr(prep.reward32), _lib.ptr(prep.advantage),
                    _lib.ptr(prep.group_tokens), _lib.ptr(prep.num_labels), _lib.ptr(prep.overflow),
                    int(per_token_columns), int(eos_token_id),
                    _lib.ptr(out["input_ids"]), _lib.ptr(out["labels"]), _lib.ptr(out["attention_mask"]),
                    _lib.ptr(out["position_ids"]), _lib.ptr(out["segment_ids"]), _lib.ptr(out["rewards"]),
                    _lib.ptr(out["advantages"]), _lib.ptr(out["ref_logprobs"]), _lib.ptr(out["old_logprobs"]),
                    _lib.ptr(out["group_tokens"]), _lib.ptr(out["num_labels"]), _lib.ptr(out["overflow"]),
                    _lib.current_stream_ptr(dev),
                )
            )
    # model_version of a micro-batch = min over its real sequences (data.py:279)
    mv = r.host_model_version
    n_mb = len(micro_batches)
    if n_mb:
        big = np.iinfo(np.int64).max
        real = pk_src >= 0
        if len(pk_src) and np.all(np.diff(mb_off) > 0):  # entries lie grouped by micro-batch: one segmented minimum (ufunc.at is ~20 x slower)
            versions = np.minimum.reduceat(np.where(real, mv[np.maximum(pk_src, 0)], big), mb_off[:-1])
        else:  # an empty micro-batch in the plan
            mb_of = np.repeat(np.arange(n_mb), np.diff(mb_off))
            versions = np.full(n_mb, big, dtype=np.int64)
            np.minimum.at(versions, mb_of[real], mv[pk_src[real]])
        versions[versions == big] = 0
    else:
        versions = np.zeros(0, dtype=np.int64)
    pads = None if sentinel_pad is None else np.asarray(sentinel_pad, dtype=np.int64)
    return PackedStep(out, pk_dst, mb_off, versions, pads)


@dataclass
class CompactBatch:
    """One packed micro-batch BEFORE expansion: the ragged columns of its sequences in packing order, on the host - the payload
    of the compact `training_data` wire (`batch_codec` kind PRLCMP01, 12-16 bytes per token instead of the 68 of a
    `PipelineBatchEncoding`).  The reference ships the expanded batch (preprocess.py:356-367 writes what `collate_packed`
    returned); here the pack kernel (K6, data.py:215-283) can run where the batch is consumed: `to_batch(device)` uploads the
    columns in one copy and launches it on the learner's GPU - the same kernel on the same inputs the preprocessor would have
    given it, so the batch is identical to the last bit (tests/test_gpu_compact_wire.py)."""

    tokens: np.ndarray               # int32 [n]
    labels: np.ndarray               # int32 [n]
    logprobs: np.ndarray             # fp32 [nc]  completion tokens only, right-aligned inside a sequence
    ref_logprobs: np.ndarray | None  # fp32 [nc]  (None: the KL term is off, `ref_logprobs` := `old_logprobs`)
    seq_off: np.ndarray              # int64 [m + 1]
    lp_off: np.ndarray               # int64 [m + 1]
    seq_scalars: np.ndarray          # fp32 [5, m]: rewards, advantages, group_tokens, num_labels, overflow per sequence
    model_version: int = 0
    padding: int = 0                 # sequence-parallel filler tokens appended by the pack kernel ((-n) % seq_parallel)
    eos_token_id: int = 0
    ref_column: np.ndarray | None = None  # fp32 [n + padding]: the EXPANDED batch's `ref_logprobs`, written by a reference policy in the preprocessor (KL on)
    slice_index: int = 0             # sequence parallelism: every rank of the group receives the whole record and keeps token slice
    num_slices: int = 1              # `slice_index` of `num_slices` after expansion (types.py:145-180 `make_slices`)

    @property
    def n_tokens(self) -> int:
        return int(self.tokens.shape[0])

    @property
    def n_seqs(self) -> int:
        return int(self.seq_off.shape[0]) - 1

    def host_facts(self) -> dict[str, Any]:
        """What `finetune_loop.annotate_host_batch` derives from an expanded batch on the host, from the ragged columns: the
        real-token count and the flat indices of the logits rows that predict a labelled token (the first token of every
        sequence but the first carries no label in the packed batch, data.py:264-265)."""
        lab = self.labels.copy()
        first = self.seq_off[1:-1]
        lab[first[first < len(lab)]] = MASKED_TOKEN_ID
        total = self.n_tokens + int(self.padding)
        if self.num_slices > 1:  # the facts of THIS rank's slice, as `annotate_host_batch` finds them on the sliced batch (filler tokens carry no label)
            lab = np.concatenate([lab, np.full(int(self.padding), MASKED_TOKEN_ID, dtype=lab.dtype)])
            lo, hi = self.slice_index * total // self.num_slices, (self.slice_index + 1) * total // self.num_slices
            lab, total = lab[lo:hi], hi - lo
        live = np.flatnonzero(lab[1:] != MASKED_TOKEN_ID)
        return {"tokens": total, "labelled_rows": torch.from_numpy(live.astype(np.int64))}

    def to_batch(self, device: torch.device | str, stager: Any = None) -> PipelineBatchEncoding:
        """Upload (ONE copy through `stager`, a `staging.PinnedStager`, when given) + one K6 launch on `device`'s current
        stream -> the packed `PipelineBatchEncoding` on that device."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("a compact micro-batch is expanded by the pack kernel: it needs a HIP device; there is no CPU fallback")
        m, n = self.n_seqs, self.n_tokens
        seq_off = np.ascontiguousarray(self.seq_off, dtype=np.int64)
        # the launch plan of ONE micro-batch whose sequences already lie in packing order: source = segment = 0 .. m - 1, destination
        # offsets = the sequence offsets themselves - it rides along with the columns (one copy), nothing is planned
        order = np.arange(m, dtype=np.int32)
        cols = [self.tokens, self.labels, self.logprobs, self.ref_logprobs, seq_off, self.lp_off, self.seq_scalars, order, self.ref_column]
        if stager is not None:
            up = stager.upload(cols)
        else:
            up = [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True) for a in cols]
        tokens, labels, lp, ref, d_off, lp_off, sc, d_order, ref_column = up

        def finish(batch: PipelineBatchEncoding) -> PipelineBatchEncoding:
            if ref_column is not None:  # the column a reference policy wrote in the preprocessor replaces what the pack kernel derived from the rollouts
                if ref_column.numel() != batch.input_ids.numel():
                    raise ValueError(f"ref_column has {ref_column.numel()} entries, the expanded batch {batch.input_ids.numel()} tokens")
                batch.ref_logprobs = ref_column.unsqueeze(0)
            if self.num_slices > 1:
                batch = batch.make_slices(self.num_slices)[self.slice_index]
            return batch

        if self.padding:  # sequence-parallel filler: the general planner knows how to append it
            none = torch.empty(0, device=dev)
            rag = RaggedRollouts(
                tokens=tokens, labels=labels, logprobs=lp, ref_logprobs=ref, seq_off=d_off, lp_off=lp_off, reward=none, group_index=none,
                step_index=none, rollout_index=none, model_version=none, finished=none, finish_code=none,
                host_seq_off=seq_off, host_lp_off=np.asarray(self.lp_off, dtype=np.int64), host_model_version=np.full(m, int(self.model_version), dtype=np.int64),
            )
            prep = PreparedRollouts(rollouts=rag, reward32=sc[0], advantage=sc[1], group_tokens=sc[2], num_labels=sc[3], overflow=sc[4],
                                    advantage64=none, group_tokens64=none)
            return finish(pack_prepared(prep, [range(m)], self.eos_token_id, sentinel_pad=[int(self.padding)], stager=stager)[0])
        out = _alloc_outputs(n, dev, packed=True)
        out.pop("__block__")
        if m and n:
            p = _lib.ptr
            with torch.cuda.device(dev):
                _lib.check(_lib.load().prl_pack_collate(
                    m, n, p(d_order), p(d_off), p(d_order), p(tokens), p(labels), p(lp), p(ref), p(d_off), p(lp_off), p(sc[0]), p(sc[1]), p(sc[2]),
                    p(sc[3]), p(sc[4]), 0, int(self.eos_token_id), p(out["input_ids"]), p(out["labels"]), p(out["attention_mask"]), p(out["position_ids"]),
                    p(out["segment_ids"]), p(out["rewards"]), p(out["advantages"]), p(out["ref_logprobs"]), p(out["old_logprobs"]), p(out["group_tokens"]),
                    p(out["num_labels"]), p(out["overflow"]), _lib.current_stream_ptr(dev)))
        return finish(PipelineBatchEncoding(**{k: v.unsqueeze(0) for k, v in out.items()}, model_version=int(self.model_version), is_packed=True,
                                            seq_boundaries=torch.from_numpy(seq_off.astype(np.int32)), padding=0))


def compact_micro_batch(host_chunks: Sequence[RaggedRollouts], seq_scalars: Sequence[np.ndarray], members: Sequence[tuple[int, int]],
                        eos_token_id: int, padding: int = 0) -> CompactBatch:
    """The compact form of the micro-batch whose sequences are `members` = (chunk, index into that chunk) in packing order:
    `host_chunks[c]` the chunk's rollouts on the host, `seq_scalars[c]` its fp32 [5, S_c] per-sequence columns (rewards + the
    four K5 outputs).  The plain-numpy statement of what the preprocessor's publisher gathers piece by piece."""
    toks, labs, lps, refs, lens, lp_lens, scal, versions = [], [], [], [], [], [], [], []
    any_ref = any(host_chunks[c].ref_logprobs is not None for c, _ in members)
    for c, i in members:
        r = host_chunks[c]
        a, b = int(r.host_seq_off[i]), int(r.host_seq_off[i + 1])
        la, lb = int(r.host_lp_off[i]), int(r.host_lp_off[i + 1])
        toks.append(r.tokens.numpy()[a:b])
        labs.append(r.labels.numpy()[a:b])
        lps.append(r.logprobs.numpy()[la:lb])
        if any_ref:  # a chunk without reference log-probs contributes its rollout log-probs, like the pack kernel does
            refs.append((r.ref_logprobs if r.ref_logprobs is not None else r.logprobs).numpy()[la:lb])
        lens.append(b - a)
        lp_lens.append(lb - la)
        scal.append(seq_scalars[c][:, i])
        versions.append(int(r.host_model_version[i]))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt, copy=False) if xs else np.zeros(0, dtype=dt)  # noqa: E731
    off = lambda ls: np.concatenate([[0], np.cumsum(ls)]).astype(np.int64)  # noqa: E731
    return CompactBatch(
        tokens=cat(toks, np.int32), labels=cat(labs, np.int32), logprobs=cat(lps, np.float32), ref_logprobs=cat(refs, np.float32) if any_ref else None,
        seq_off=off(lens), lp_off=off(lp_lens), seq_scalars=np.ascontiguousarray(np.stack(scal, axis=1), dtype=np.float32) if scal else np.zeros((5, 0), np.float32),
        model_version=min(versions) if versions else 0, padding=padding, eos_token_id=eos_token_id,
    )


def pad_prepared(
    prep: PreparedRollouts,
    rows: Sequence[int],
    padding_side: str = "right",
    pad_to_multiple_of: int = 16,
    per_token_columns: int = 0,
) -> PipelineBatchEncoding:
    """K7: one padded [B, Lp] batch out of sequences `rows` (reference data.py:163-212)."""
    lib = _lib.load()
    r = prep.rollouts
    dev = r.device
    lens = r.seq_lengths()
    longest = int(max(lens[s] for s in rows))
    if longest % pad_to_multiple_of:
        longest += pad_to_multiple_of - (longest % pad_to_multiple_of)
    B = len(rows)
    out = _alloc_outputs(B * longest, dev, packed=False)
    d_rows = torch.from_numpy(np.asarray(list(rows), dtype=np.int32)).to(dev, non_blocking=True)
    with torch.cuda.device(dev):
        _lib.check(
            lib.prl_pad_collate(
                B, longest, int(padding_side != "right"), _lib.ptr(d_rows), _lib.ptr(r.tokens), _lib.ptr(r.labels),
                _lib.ptr(r.logprobs), _lib.ptr(r.ref_logprobs), _lib.ptr(r.seq_off), _lib.ptr(r.lp_off),
                _lib.ptr(prep.reward32), _lib.ptr(prep.advantage), _lib.ptr(prep.group_tokens),
                _lib.ptr(prep.num_labels), _lib.ptr(prep.overflow), int(per_token_columns),
                _lib.ptr(out["input_ids"]), _lib.ptr(out["labels"]), _lib.ptr(out["attention_mask"]),
                _lib.ptr(out["rewards"]), _lib.ptr(out["advantages"]), _lib.ptr(out["ref_logprobs"]),
                _lib.ptr(out["old_logprobs"]), _lib.ptr(out["group_tokens"]), _lib.ptr(out["num_labels"]),
                _lib.ptr(out["overflow"]), _lib.current_stream_ptr(dev),
            )
        )
    fields = {k: v.view(B, longest) for k, v in out.items() if k != "__block__"}
    mv = r.host_model_version
    return PipelineBatchEncoding(**fields, model_version=int(min(mv[list(rows)])), is_packed=False)


# ---------------------------------------------------------------------------------------------
# list-of-dicts level (reference signatures)
# ---------------------------------------------------------------------------------------------


def _examples_to_prepared(examples: Sequence[dict[str, Any]], device: torch.device) -> tuple[PreparedRollouts, int]:
    """Flatten example dicts (per-token python lists) into ragged device buffers.  All seven RL
    columns are taken per token (`per_token_columns` = all bits) so arbitrary user-built lists
    survive unchanged; old/ref logprobs are passed full-length (no implicit left padding)."""
    n = len(examples)
    lens = np.fromiter((len(e["input_ids"]) for e in examples), dtype=np.int64, count=n)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])

    def flat(key: str, dtype, default=0) -> np.ndarray:
        buf = np.empty(total, dtype=dtype)
        for i, e in enumerate(examples):
            v = e.get(key)
            if v is None:
                buf[off[i] : off[i + 1]] = default
            else:
                if len(v) != lens[i]:
                    raise ValueError(f"example {i}: column {key!r} has {len(v)} entries, expected {lens[i]}")
                buf[off[i] : off[i + 1]] = v
        return buf

    dev_t = lambda a: torch.from_numpy(a).to(device, non_blocking=True)  # noqa: E731
    tokens = flat("input_ids", np.int32)
    labels = flat("labels", np.int32)
    cols = {k: flat(k, np.float32, 0.0) for k in RL_DATA_COLUMNS}
    mv = np.fromiter((e.get("model_version", 0) for e in examples), dtype=np.int64, count=n)
    zeros_i32 = np.zeros(n, dtype=np.int32)
    rag = RaggedRollouts.from_numpy(
        tokens, labels, cols["old_logprobs"], cols["ref_logprobs"], off, off, np.zeros(n), zeros_i32, zeros_i32,
        zeros_i32, mv, np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8),
    ).to(device)
    prep = PreparedRollouts(
        rollouts=rag, reward32=dev_t(cols["rewards"]), advantage=dev_t(cols["advantages"]),
        group_tokens=dev_t(cols["group_tokens"]), num_labels=dev_t(cols["num_labels"]),
        overflow=dev_t(cols["overflow"]), advantage64=torch.empty(0), group_tokens64=torch.empty(0),
    )
    return prep, sum(_PER_TOKEN_BITS.values())


def _device_for_collate() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("pipelinerl_amd collate kernels need a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def collate_packed(
    examples: list[dict[str, list[int]]],
    tokenizer: Any,
    seq_parallel: int,
    label_pad_value: int = MASKED_TOKEN_ID,
) -> PipelineBatchEncoding:
    """Concatenate examples into one [1, T] packed batch (reference data.py:215-283).  Tensors stay
    on the HIP device."""
    if label_pad_value != MASKED_TOKEN_ID:
        raise ValueError("label_pad_value other than -100 is not supported")
    if not examples:  # the reference's first failure on an empty list is examples[0] (data.py:246): same exception, before any device work
        raise IndexError("list index out of range")
    total = sum(len(e["input_ids"]) for e in examples)
    pad = (seq_parallel - total % seq_parallel) % seq_parallel if seq_parallel > 0 else 0
    prep, bits = _examples_to_prepared(examples, _device_for_collate())
    batch = pack_prepared(
        prep, [list(range(len(examples)))], eos_token_id=getattr(tokenizer, "eos_token_id", 0) or 0,
        sentinel_pad=[pad], per_token_columns=bits,
    )[0]
    return batch


def collate(
    examples: list[dict[str, list[int]]],
    tokenizer: Any,
    label_mask_value: int = MASKED_TOKEN_ID,
    pad_to_multiple_of: int = 16,
) -> PipelineBatchEncoding:
    """Pad examples to a common length, rounded up to `pad_to_multiple_of` (reference
    data.py:163-212); padding side from `tokenizer.padding_side`."""
    if label_mask_value != MASKED_TOKEN_ID:
        raise ValueError("label_mask_value other than -100 is not supported")
    if not examples:  # the reference reads examples[0].keys() first (data.py:170)
        raise IndexError("list index out of range")
    prep, bits = _examples_to_prepared(examples, _device_for_collate())
    return pad_prepared(
        prep, list(range(len(examples))), padding_side=getattr(tokenizer, "padding_side", "right"),
        pad_to_multiple_of=pad_to_multiple_of, per_token_columns=bits,
    )


# ---------------------------------------------------------------------------------------------
# host helpers with the reference's signatures
# ---------------------------------------------------------------------------------------------


def validate_spans(text: str, predicted_spans: list[tuple[int, int]]) -> None:
    """Spans must lie inside the text, be well-formed, ordered and non-overlapping."""
    prev_end = None
    for begin, end in predicted_spans:
        if begin < 0 or end > len(text):
            raise ValueError(f"Span {begin}:{end} is out of bounds for text {text!r}")
        if begin > end:
            raise ValueError(f"Span {begin}:{end} is invalid")
        if prev_end is not None and begin < prev_end:
            raise ValueError(f"Span {begin}:{end} overlaps the previous span ending at {prev_end}")
        prev_end = end


def mask_labels(
    input_ids: Sequence[int],
    offset_mapping: Iterable[tuple[int, int]],
    predicted_spans: Iterable[Iterable[int]],
    masked_token_id: int = MASKED_TOKEN_ID,
) -> tuple[list[int], list[int]]:
    """Labels = input ids where the token's character range overlaps a predicted span, else masked;
    also the first overlapping token index per span (reference data.py:47-93)."""
    offsets = list(offset_mapping)
    labels = [masked_token_id] * len(input_ids)
    midpoints: list[int] = []
    for span in predicted_spans:
        lo, hi = tuple(span)
        first = None
        for i, (tb, te) in enumerate(offsets):
            if tb < hi and lo < te:
                labels[i] = input_ids[i]
                if first is None:
                    first = i
        if first is not None:
            midpoints.append(first)
    return labels, midpoints


def preprocess_fn(entry: dict[str, Any], tokenizer: Any, seq_length: int, is_rl: bool = False) -> dict[str, Any]:
    """One rollout record -> per-token python lists (reference data.py:111-160).  Compatibility
    helper; the device path consumes `RaggedRollouts` directly."""
    if entry.get("input_ids"):
        n = len(entry["input_ids"])
        encoding: dict[str, Any] = {
            "input_ids": entry["input_ids"],
            "labels": entry["labels"],
            "attention_mask": [1] * n,
        }
    else:
        text = entry["text"]
        encoding = dict(tokenizer(text, return_offsets_mapping=True, max_length=seq_length, truncation=True))
        if "predicted_spans" in entry:
            spans = entry["predicted_spans"]
        else:
            n_chars = entry.get("n_predicted", len(text))
            spans = [(len(text) - n_chars, len(text))]
        validate_spans(text, spans)
        encoding["labels"], _ = mask_labels(encoding["input_ids"], encoding["offset_mapping"], spans)
    if is_rl:
        encoding = prepare_rl_fields(encoding, entry["reward"], entry["logprobs"], entry["ref_logprobs"])
    for key in ("pixel_values", "image_thw"):
        if key in entry:
            encoding[key] = entry[key]
    return encoding


__all__ = [
    "MASKED_TOKEN_ID",
    "PackedStep",
    "collate",
    "collate_packed",
    "create_sentinel_example",
    "mask_labels",
    "pack_prepared",
    "pad_prepared",
    "preprocess_fn",
    "validate_spans",
]
