"""Preprocessor stage: micro-batch scheduling and the device preprocess of a chunk.

The reference's `run_preprocessing_loop` (pipelinerl/preprocess.py:370-704) interleaves queue
plumbing with the scheduling rule that decides which samples go to which trainer in which
micro-batch.  That rule (preprocess.py:462-481, 596-662) is reproduced here as a
pure state machine, `MicroBatchScheduler`, because the trainer's sample accounting asserts
depend on it (finetune_loop.py:674-675, 859); the numeric work of a chunk
(`preprocess_dataset`, preprocess.py:145-189) is `preprocess_chunk` -> K5 on device.
"""

from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import Any, Iterable, Sequence

import numpy as np

import logging
import time

import torch

from .finetune.rl import PreparedRollouts, RLConfig, populate_rl_data_ragged
from .ragged import RaggedRollouts, concat_ragged

logger = logging.getLogger(__name__)


@dataclass
class MicroBatch:
    """One scheduling decision: `samples` go to lead trainer `trainer_id` (a sentinel when empty)."""

    trainer_id: int
    samples: list[Any] = field(default_factory=list)
    sentinel: bool = False


class MicroBatchScheduler:
    """Greedy first-fit packing with per-step sample quotas and sentinel fill.

    num_lead_trainers           = num_trainers // seq_parallel
    samples_per_lead_per_step   = train_batch_size * (gradient_accumulation_passes // num_lead_trainers)
    samples_per_step            = samples_per_lead_per_step * num_lead_trainers

    Packed mode: pop samples from the head of the queue while they fit the `seq_length` token
    budget; flush when the next one would overflow or when the trainer reaches its quota for the
    step; a trainer that already holds its quota receives a sentinel micro-batch so that every
    rank runs the same number of forward/backward passes; the step closes when all quotas are
    full and the round-robin is back at trainer 0.  Unpacked mode: fixed `train_batch_size`
    samples per micro-batch, same round-robin.
    """

    def __init__(
        self,
        num_trainers: int,
        train_batch_size: int,
        gradient_accumulation_passes: int,
        seq_length: int,
        seq_parallel: int = 1,
        seq_packing: bool = True,
        published_samples: int = 0,
        length_of=len,
    ):
        if num_trainers % seq_parallel:
            raise ValueError("num_trainers must be a multiple of seq_parallel")
        self.num_trainers = num_trainers
        self.seq_parallel = seq_parallel
        self.seq_length = seq_length
        self.seq_packing = seq_packing
        self.batch_size_per_call = train_batch_size
        self.num_lead_trainers = num_trainers // seq_parallel
        passes_per_lead = gradient_accumulation_passes // self.num_lead_trainers
        self.samples_per_lead_per_step = train_batch_size * passes_per_lead
        self.samples_per_step = self.samples_per_lead_per_step * self.num_lead_trainers
        assert published_samples % self.num_lead_trainers == 0
        self.published_samples = published_samples
        self.samples_per_trainer = {i: published_samples // num_trainers for i in range(0, num_trainers, seq_parallel)}
        self.trainer_id = 0
        self.batch_boundary = published_samples + self.samples_per_step
        self.target_samples_per_lead = self.samples_per_trainer[0] + self.samples_per_lead_per_step
        self.queue: deque = deque()
        self._current: list[Any] = []
        self._current_length = 0
        self._length_of = length_of

    def push(self, samples: Iterable[Any]) -> None:
        self.queue.extend(samples)  # `queue` may be replaced by a bounded ring (PreprocessorLoop)

    def _advance(self) -> None:
        self.trainer_id = (self.trainer_id + self.seq_parallel) % self.num_trainers

    def drain(self) -> tuple[list[MicroBatch], bool]:
        """Run the scheduling loop until the queue is empty or a step's batch is complete.
        Returns (micro-batches in emission order, batch_done)."""
        out: list[MicroBatch] = []
        batch_done = False
        while self.queue and not batch_done:
            tid = self.trainer_id
            if self.seq_packing:
                if self.samples_per_trainer[tid] == self.target_samples_per_lead:
                    out.append(MicroBatch(tid, [], sentinel=True))
                    self._advance()
                else:
                    flush = False
                    while self.queue:
                        n = self._length_of(self.queue[0])
                        if self._current_length + n > self.seq_length:
                            flush = True
                            break
                        self._current.append(self.queue.popleft())
                        self._current_length += n
                        if len(self._current) + self.samples_per_trainer[tid] == self.target_samples_per_lead:
                            flush = True
                            break
                    if flush:
                        assert len(self._current) > 0, "Current batch should not be empty when writing"
                        out.append(MicroBatch(tid, self._current))
                        self.published_samples += len(self._current)
                        self.samples_per_trainer[tid] += len(self._current)
                        self._advance()
                        self._current, self._current_length = [], 0
            else:
                if len(self.queue) < self.batch_size_per_call:
                    break  # the reference would pop from an empty deque here; wait for more data
                picked = [self.queue.popleft() for _ in range(self.batch_size_per_call)]
                out.append(MicroBatch(tid, picked))
                self.published_samples += len(picked)
                self.samples_per_trainer[tid] += len(picked)
                self._advance()
            batch_done = self.published_samples == self.batch_boundary and self.trainer_id == 0
            if batch_done:
                self.batch_boundary += self.samples_per_step
                self.target_samples_per_lead += self.samples_per_lead_per_step
        return out, batch_done


def filter_zero_advantage_groups(dataset: list[dict], epsilon: float = 1e-6) -> tuple[list[dict], int]:
    """Drop every group whose advantages are all (near) zero (reference preprocess.py:316-353).
    Returns (kept entries in group order of first appearance, number dropped)."""
    by_group: dict[Any, list[dict]] = {}
    for e in dataset:
        by_group.setdefault(e["group_id"], []).append(e)
    kept: list[dict] = []
    dropped = 0
    for entries in by_group.values():
        if any(abs(a) > epsilon for e in entries for a in e["advantages"]):
            kept.extend(entries)
        else:
            dropped += len(entries)
    return kept, dropped


def nonzero_advantage_mask(prep: PreparedRollouts, epsilon: float = 1e-6) -> np.ndarray:
    """Device-path version of the filter: boolean [S] mask of sequences whose GROUP has any
    |advantage| > epsilon."""
    adv = prep.advantage64.abs().cpu().numpy()
    gi = prep.rollouts.host_group_index
    alive = np.zeros(int(gi.max()) + 1 if len(gi) else 0, dtype=bool)
    np.logical_or.at(alive, gi, adv > epsilon)
    return alive[gi]


def check_group_sizes(entries: Sequence[dict], group_size: int) -> bool:
    """Every group_id must appear with exactly `group_size` distinct rollout_index values
    (reference preprocess.py:70-83)."""
    seen: dict[Any, set] = {}
    for e in entries:
        meta = e.get("metadata") or {}
        seen.setdefault(e["group_id"], set()).add(meta.get("rollout_index", e.get("rollout_index")))
    return all(len(v) == group_size for v in seen.values())


def preprocess_chunk(entries: Sequence[dict], eos_token_id: int, rl_config: RLConfig, device) -> PreparedRollouts:
    """`preprocess_dataset` for a chunk of `actor`-stream records: flatten to ragged SoA, upload,
    K5 on device.  `ref_logprobs` default to the rollout logprobs when the KL term is off
    (reference preprocess.py:160-161)."""
    rag = RaggedRollouts.from_entries(entries).to(device)
    return populate_rl_data_ragged(rag, eos_token_id, rl_config)


def concat_prepared(parts: Sequence[PreparedRollouts]) -> PreparedRollouts:
    """Merge preprocessed chunks (a micro-batch may span chunk boundaries)."""
    if len(parts) == 1:
        return parts[0]
    cat = lambda name: torch.cat([getattr(p, name) for p in parts])  # noqa: E731
    return PreparedRollouts(
        rollouts=concat_ragged([p.rollouts for p in parts]), reward32=cat("reward32"), advantage=cat("advantage"),
        group_tokens=cat("group_tokens"), num_labels=cat("num_labels"), overflow=cat("overflow"),
        advantage64=cat("advantage64"), group_tokens64=cat("group_tokens64"),
    )


# ---------------------------------------------------------------------------------------------
# the lossy / elastic half of the loop (reference preprocess.py:190-282, 565-585, 664-694)
# ---------------------------------------------------------------------------------------------


class ChunkLoader:
    """Reader thread of the preprocessor (`run_dataset_loader`, reference :190-228): takes
    `chunk_n_groups` groups at a time from the input stream, checks the group sizes, and hands the chunk
    to a BOUNDED queue.  When the queue is full and `pop_old_data` is on, the OLDEST waiting chunk is
    dropped to make room (the learner should train on fresh rollouts rather than stall the actor);
    otherwise the put blocks.  An error ends the thread after being forwarded through the queue.
    A group is a list of TrainingText dicts (text record) or a `RaggedRollouts` (binary record)."""

    def __init__(self, raw_chunk_queue, data_stream, check_group_size: int, chunk_n_groups: int, pop_old_data: bool,
                 reader_factory=None):
        self.queue = raw_chunk_queue
        self.data_stream = data_stream
        self.check_group_size = check_group_size
        self.chunk_n_groups = chunk_n_groups
        self.pop_old_data = pop_old_data
        self.old_and_dropped = 0
        self._reader_factory = reader_factory

    @staticmethod
    def group_sizes_ok(groups: list, group_size: int) -> bool:
        for g in groups:
            if isinstance(g, RaggedRollouts):
                pairs = np.unique(np.stack([g.host_group_index, g.host_rollout_index]), axis=1)
                if not (np.bincount(pairs[0]) == group_size).all():
                    return False
        texts = [e for g in groups if not isinstance(g, RaggedRollouts) for e in g]
        return check_group_sizes(texts, group_size)

    def run(self) -> None:
        import queue as _q

        from .streams import read_stream

        last_notice = 0
        with (self._reader_factory or read_stream)(self.data_stream) as reader:
            while True:
                try:
                    groups = []
                    for group in reader.read():
                        groups.append(group)
                        if len(groups) == self.chunk_n_groups:
                            break
                    if not self.group_sizes_ok(groups, self.check_group_size):
                        raise ValueError("Invalid group sizes in data")
                    try:
                        self.queue.put_nowait(groups)
                    except _q.Full:
                        if self.pop_old_data:
                            try:
                                self.queue.get_nowait()
                                self.old_and_dropped += 1
                                if self.old_and_dropped // 100 != last_notice:
                                    logger.info(f"So far removed {self.old_and_dropped} old elements from preprocessor queue")
                                    last_notice = self.old_and_dropped // 100
                            except _q.Empty:
                                pass
                        self.queue.put(groups)  # blocking; after a drop there is room in most cases
                except Exception as e:  # noqa: BLE001 - forwarded to the main loop
                    logger.error(f"Error in dataset loader: {e}")
                    self.queue.put(e)
                    break


class SlidingWindowAggregator:
    """Samples / tokens per second over the last `window_size` updates (reference :239-282)."""

    def __init__(self, window_size: int, clock=time.time):
        self.window_size = window_size
        self.tokens_window: list[list[int]] = []
        self.timestamps: list[float] = []
        self._clock = clock

    def has_enough_data(self) -> bool:
        return len(self.tokens_window) == self.window_size

    def update(self, token_counts: list[int]) -> None:
        self.tokens_window.append(list(token_counts))
        self.timestamps.append(self._clock())
        if len(self.tokens_window) > self.window_size:
            del self.tokens_window[0], self.timestamps[0]

    def get_stats(self) -> dict[str, float]:
        span = self.timestamps[-1] - self.timestamps[0] if self.timestamps else 0.0
        if span < 1e-6:
            return {"samples_per_second": 0, "tokens_per_second": 0}
        return {"samples_per_second": sum(len(t) for t in self.tokens_window) / span,
                "tokens_per_second": sum(sum(t) for t in self.tokens_window) / span}


class _TrackedDeque(deque):
    """`deque(maxlen)` of samples that keeps their lengths and model versions in two parallel deques, so that the
    per-admission bookkeeping of `ProcessedRing.admit` (the reference recomputes both lists over the whole ring for
    every admitted sample, :572-585) is two C-level passes instead of two Python loops.  Only the operations the loop
    and the scheduler use are tracked: append, popleft, extend, clear."""

    def __init__(self, maxlen: int, length_of, version_of):
        super().__init__(maxlen=maxlen)
        self.lengths: deque = deque(maxlen=maxlen)
        self.versions: deque = deque(maxlen=maxlen)
        self._length_of, self._version_of = length_of, version_of

    def append(self, x) -> None:
        super().append(x)
        self.lengths.append(self._length_of(x))
        self.versions.append(self._version_of(x))

    def extend(self, xs) -> None:
        for x in xs:
            self.append(x)

    def popleft(self):
        self.lengths.popleft()
        self.versions.popleft()
        return super().popleft()

    def clear(self) -> None:
        super().clear()
        self.lengths.clear()
        self.versions.clear()


class ProcessedRing:
    """The ring of preprocessed samples the scheduler draws from (`processed_entries_queue`, a
    `deque(maxlen=ring_buffer_size)`, reference :455, :572-585).  `admit` moves samples from the arrival
    buffer into the ring: with `pop_old_data` a full ring drops its OLDEST sample for every new one;
    without, admission stops until the scheduler has made room.  After every admitted sample the
    throughput window is updated with the lengths of everything in the ring and `max_model_version`
    becomes the newest model version in it - exactly the reference's bookkeeping."""

    def __init__(self, maxlen: int, pop_old_data: bool, stats: SlidingWindowAggregator | None = None,
                 length_of=lambda s: s.length, version_of=lambda s: s.model_version):
        self.entries: deque = _TrackedDeque(maxlen, length_of, version_of)
        self.pop_old_data = pop_old_data
        self.popped = 0
        self.max_model_version: int | None = None
        self.stats = stats
        self._length_of, self._version_of = length_of, version_of

    def admit(self, buffer: deque) -> None:
        q = self.entries
        while buffer:
            if len(q) == q.maxlen:
                if not self.pop_old_data:
                    break
                self.popped += 1
                if self.popped % 100 == 0:
                    logger.warning(f"Popped {self.popped} old entries from processed entries queue")
            q.append(buffer.popleft())  # a full deque(maxlen) drops from the left
            if self.stats is not None:
                self.stats.update(q.lengths)
            self.max_model_version = max(q.versions, default=0)


def preprocessor_stats_record(published_samples: int, max_model_version: Any, raw_queue_chunks: int, output_queue_chunks: int,
                              chunk_n_groups: int, attempts: int, num_filtered_out: int, total_filtered_out: int,
                              aggregator: SlidingWindowAggregator | None) -> dict[str, Any]:
    """One record of the `preprocessor_stats` stream, keys and arithmetic of reference :669-681."""
    per_chunk = chunk_n_groups * attempts
    stats = {
        "preprocessor/published_samples": published_samples,
        "preprocessor/published_model_version": max_model_version,
        "preprocessor/queue/raw_samples": raw_queue_chunks * per_chunk,
        "preprocessor/queue/raw": raw_queue_chunks,
        "preprocessor/queue/output_samples": output_queue_chunks * per_chunk,
        "preprocessor/queue/output": output_queue_chunks,
        "preprocessor/filtered_out_samples": num_filtered_out,
        "preprocessor/total_filtered_out_samples": total_filtered_out,
    }
    if aggregator is not None and aggregator.has_enough_data():
        stats.update({"preprocessor/" + k: v for k, v in aggregator.get_stats().items()})
    return stats


def should_write_stats(published_samples: int, last_published_samples: int, debug_mode: Any, batch_done: bool, log_every_n_samples: int) -> bool:
    """When the reference emits a stats record (:664-667)."""
    return published_samples > last_published_samples and bool(
        debug_mode or batch_done or (published_samples - last_published_samples > log_every_n_samples))


def replace_oov_tokens_with_the(data: list[dict], tokenizer: Any) -> list[dict]:
    """Host front end with the reference's contract (:107-141): token ids that are not in the
    tokenizer's vocabulary become the id of "the" in `input_ids` (labels are left as they are)."""
    vocab = tokenizer.get_vocab()
    valid = np.zeros(max(vocab.values()) + 1, dtype=bool)
    valid[list(vocab.values())] = True
    the_id = vocab["the"]
    patched = 0
    for entry in data:
        ids = np.asarray(entry["input_ids"], dtype=np.int64)
        ok = (ids >= 0) & (ids < len(valid))
        ok[ok] = valid[ids[ok]]
        if not ok.all():
            patched += 1
            logger.warning(f"Patching entry with invalid token ids: {ids[~ok].tolist()}")
            entry["input_ids"] = np.where(ok, ids, the_id).tolist()
    if patched:
        logger.warning(f"Patched {patched} entries with invalid token ids from {len(data)}")
    return data


class OovPatcher:
    """Device form of the above for ragged rollouts: one pass of `prl_patch_oov` over the token buffer."""

    def __init__(self, vocab_ids: Iterable[int], the_token_id: int, device):
        ids = np.fromiter(vocab_ids, dtype=np.int64)
        table = np.zeros(int(ids.max()) + 1 if len(ids) else 0, dtype=np.uint8)
        table[ids] = 1
        self.valid = torch.from_numpy(table).to(device)
        self.the_token_id = int(the_token_id)
        self.count = torch.zeros(1, dtype=torch.int64, device=device)

    @classmethod
    def from_tokenizer(cls, tokenizer: Any, device) -> "OovPatcher":
        vocab = tokenizer.get_vocab()
        return cls(vocab.values(), vocab["the"], device)

    def apply(self, rollouts: RaggedRollouts) -> RaggedRollouts:
        from . import _lib

        t = rollouts.tokens
        _lib.require_device(t)
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().prl_patch_oov(t.numel(), t.data_ptr(), self.valid.data_ptr(), self.valid.numel(), self.the_token_id,
                                                 self.count.data_ptr(), _lib.current_stream_ptr(t.device)))
        return rollouts


@dataclass
class PreprocessorConfig:
    """The keys of the reference config the preprocessor loop consumes (conf/base.yaml:25-44, 105,
    conf/finetune/base.yaml; SURVEY.md §5 "config"), same defaults."""

    exp_path: Any
    num_trainers: int
    train_batch_size: int
    gradient_accumulation_passes: int
    seq_length: int
    attempts: int
    rl: RLConfig
    eos_token_id: int
    seq_parallel: int = 1
    seq_packing: bool = True
    padding_side: str = "right"
    chunk_n_groups: int = 2
    raw_queue_size: int = 8
    dataset_buffer_size: int = 0
    ring_buffer_size: int = 128
    max_ready_samples_per_lead: int = 64
    pop_old_data: bool = True
    max_lag: int | None = None
    debug_mode: Any = None
    log_every_n_samples: int = 128
    samples_target: int | None = None  # final_train_steps * train_batch_size * gradient_accumulation_passes (:431-432)
    input_topic: str = "actor"
    output_topic: str = "training_data"
    stats_topic: str = "preprocessor_stats"

    @property
    def drops_old_data(self) -> bool:
        """`pop_old_data = cfg.max_lag is None and cfg.pop_old_data and not cfg.debug.mode` (:408)."""
        return self.max_lag is None and self.pop_old_data and not self.debug_mode


_PUB_RECORD_DT = np.dtype([("log", "<u8"), ("nbytes", "<u8"), ("first_piece", "<u4"), ("n_pieces", "<u4")])  # prl_pub_record
_PUB_PIECE_DT = np.dtype([("src", "<u8"), ("offset", "<u8"), ("nbytes", "<u8"), ("kind", "<u4"), ("_pad", "<u4")])  # prl_pub_piece


def compact_sources(rollouts: RaggedRollouts, k5_out32: np.ndarray) -> dict:
    """Where the compact wire gathers a chunk's columns from: base addresses of the HOST arrays of its decoded `actor` records,
    its offsets, and the fp32 [5, S] per-sequence scalars in the order of `CompactBatch.seq_scalars` - rewards from the host record
    (the same fp64 -> fp32 rounding the device applies) + K5's outputs (`k5_out32` rows: num_labels, overflow, advantage, group_tokens)."""
    r = rollouts
    scal = np.empty((5, r.n_seqs), dtype=np.float32)
    scal[0] = r.reward.numpy().astype(np.float32)
    scal[1], scal[2], scal[3], scal[4] = k5_out32[2], k5_out32[3], k5_out32[0], k5_out32[1]
    ref = r.ref_logprobs if r.ref_logprobs is not None else r.logprobs  # like the pack kernel: no reference log-probs = the rollout's own
    return dict(rollouts=r, scalars=scal, tok=r.tokens.data_ptr(), lab=r.labels.data_ptr(), lp=r.logprobs.data_ptr(), ref=ref.data_ptr(),
                has_ref=r.ref_logprobs is not None, seq_off=r.host_seq_off, lp_off=r.host_lp_off)


def describe_compact(members: Sequence[tuple[dict, int]], model_version: int, eos_token_id: int, inline: bytearray, pieces: list,
                     padding: int = 0, slice_index: int = 0, num_slices: int = 1, ref_block: tuple[int, int] | None = None) -> int:
    """(`padding`, `slice_index` / `num_slices`: the sequence-parallel form - the record names the filler the pack kernel appends and the
    token slice its reader keeps.  `ref_block` = (byte offset, nbytes) of the micro-batch's `ref_logprobs` column inside the job's
    device block: one PRL_PUB_FROM_BLOCK piece, the column a reference policy wrote in the preprocessor.)  The PRLCMP01 record of the micro-batch whose sequences are `members` = (chunk sources, index in that chunk) in packing
    order, as a recipe for the native publisher: appends `prl_pub_piece` rows (src, offset in the record, nbytes, kind, 0) to
    `pieces` - one PRL_PUB_FROM_HOST piece per sequence and per-token column, header and per-sequence arrays PRL_PUB_INLINE in
    `inline` - and returns the record size.  Byte for byte `batch_codec.encode_compact(finetune.data.compact_micro_batch(...))`."""
    from . import _lib, batch_codec

    FROM_HOST, INLINE, FROM_BLOCK = _lib.PRL_PUB_FROM_HOST, _lib.PRL_PUB_INLINE, _lib.PRL_PUB_FROM_BLOCK
    spans = [(int(hc["seq_off"][i]), int(hc["seq_off"][i + 1]), int(hc["lp_off"][i]), int(hc["lp_off"][i + 1])) for hc, i in members]
    lens = [b - a for a, b, _, _ in spans]
    lp_lens = [lb - la for _, _, la, lb in spans]
    m = len(members)
    has_ref = any(hc["has_ref"] for hc, _ in members)
    head, base, where, total = batch_codec.compact_layout(sum(lens), sum(lp_lens), m, has_ref, model_version, padding, eos_token_id,
                                                          slice_index, num_slices, (ref_block[1] // 4) if ref_block else 0)
    if ref_block is not None and ref_block[1] != 4 * (sum(lens) + padding):
        raise ValueError(f"ref column of {ref_block[1]} bytes for a micro-batch of {sum(lens)} + {padding} tokens")
    at = len(inline)
    inline += head
    pieces.append((at, 0, len(head), INLINE, 0))
    for col, key, width in (("tokens", "tok", 0), ("labels", "lab", 0), ("logprobs", "lp", 2), ("ref_logprobs", "ref", 2)):
        if col == "ref_logprobs" and not has_ref:
            continue
        off = base + where[col][0]
        for (hc, _), span in zip(members, spans):
            a, b = span[width], span[width + 1]
            if b > a:
                pieces.append((hc[key] + 4 * a, off, 4 * (b - a), FROM_HOST, 0))
                off += 4 * (b - a)
    from itertools import accumulate

    seq_off = np.array([0, *accumulate(lens)], dtype=np.int64).tobytes()
    lp_off = np.array([0, *accumulate(lp_lens)], dtype=np.int64).tobytes()
    if m == 1:
        hc, i = members[0]
        scal = hc["scalars"][:, i].tobytes()  # [5, 1] row-major = the column itself
    else:
        scal = np.ascontiguousarray(np.stack([hc["scalars"][:, i] for hc, i in members], axis=1), dtype=np.float32).tobytes() if m else b""
    for col, raw in (("seq_off", seq_off), ("lp_off", lp_off), ("seq_scalars", scal)):
        if raw:
            at = len(inline)
            inline += raw
            pieces.append((at, base + where[col][0], len(raw), INLINE, 0))
    if ref_block is not None:  # the last column of the record (the log gathers a record's pieces in ascending offset)
        pieces.append((ref_block[0], base + where["ref_column"][0], ref_block[1], FROM_BLOCK, 0))
    return total


@dataclass
class _Sample:
    chunk: int
    index: int
    length: int
    model_version: int = 0


class PreprocessorLoop:
    """`run_preprocessing_loop` of the reference (preprocess.py:370-704) without worker processes: the
    per-chunk work that needed N CPU workers is two kernel launches here.

        actor stream -> ChunkLoader (bounded queue, drop-oldest) -> K5 on device (+ OOV patch, zero-advantage
        filter) -> arrival buffer -> ProcessedRing (drop-oldest) -> MicroBatchScheduler -> ONE K6 launch per
        drain (packed, with sequence-parallel fillers) or K7 (unpacked) -> per-trainer `training_data`
        partitions (+ sentinels), `preprocessor_stats` records

    Back-pressure as in the reference (:587-592): publishing pauses while
    published - trainer_state.samples_processed exceeds max_ready_samples_per_lead * num_trainers."""

    def __init__(self, cfg: PreprocessorConfig, device, trainer_state=None, ref_model=None, oov_patcher: OovPatcher | None = None,
                 batched_transfers: bool = True, profile: bool = False, overlap_publish: bool = True, wire: str = "full"):
        """`ref_model`: a frozen reference policy on the preprocessor's GPU.  When given, every real
        micro-batch gets its `ref_logprobs` from a no-grad forward of that model (hidden states -> MFMA head for a
        model in the Hugging Face layout, K1 on its logits otherwise)
        before it is published - the device-side replacement of the reference's HTTP round trip to a
        second inference server (preprocess.py:86-104, llm.py:606-648).
        `batched_transfers` (default): a chunk crosses the bus in ONE host -> device copy (13 ragged arrays + the K5 plan
        through a page-locked ring, `staging.PinnedStager`), the K6 plan in one more, and a drain's packed micro-batches
        come back in ONE device -> host copy before they are encoded into stream records; off = one copy per array
        (the round-3 form, kept for A/B).
        `overlap_publish` (default; shm backend, packed mode, on a GPU): the device -> host copy of a drain's packed block and the
        gathering of its records into the trainers' logs run on the NATIVE publisher thread of libprl (csrc/prl_publish.cpp: its
        own copy stream and page-locked double buffer), in drain order, at most two drains deep - chunk k leaves the device and
        enters the log while chunk k + 1 is ingested, scanned (K5) and packed (K6).  The reference publishes inline
        (preprocess.py:356-367, 629-648); what a reader of `training_data` sees is the same record sequence, byte for byte.
        Off, another backend, a JSONL mirror, unpacked mode: publish inline.
        `wire`: what a `training_data` record holds.  "full" (default) = the reference's record, the expanded
        `PipelineBatchEncoding` (68 bytes per token: K6 runs here, the packed block comes back over the bus and into the log).
        "compact" = the micro-batch BEFORE expansion (`batch_codec` kind PRLCMP01, 12-16 bytes per token): the ragged columns of
        its sequences, gathered by the publisher straight from the decoded `actor` records on the host, + the five per-sequence
        scalars (K5's outputs: 16 bytes per SEQUENCE come back from the device); K6 then runs on the learner's GPU
        (`finetune.data.CompactBatch.to_batch`, called by `finetune_loop.run_data_loader`) and produces the identical batch.
        Needs the shm backend and packing.  Sequence parallelism: every rank of an SP group gets the whole record (with the filler
        count) and keeps its `make_slices` slice after expansion (types.py:145-180).  A reference policy here (`ref_model`, KL on):
        K6 and its forward run on THIS GPU as on the full wire, and of the packed block only the `ref_logprobs` column - 4 bytes per
        token - comes back and rides in the record (`ref_column`).  `oov_patcher` rewrites token ids on the device: the chunk's patched
        ids are copied back into its host record (4 bytes per token) before anything is gathered from it.
        `profile`: accumulate host wall time per phase in `self.prof` (seconds; `perf_counter` pairs, no device sync)."""
        from .streams import SingleStreamSpec, StreamRangeSpec

        if wire not in ("full", "compact"):
            raise ValueError(f"wire must be 'full' or 'compact', not {wire!r}")
        if wire == "compact":
            if not cfg.seq_packing:
                raise ValueError("the compact wire carries packed micro-batches (seq_packing=True)")
            if torch.device(device).type != "cuda":
                raise RuntimeError("the preprocessor's kernels need a HIP device; there is no CPU fallback")
        self.wire = wire
        self.host_chunks: dict[int, dict] = {}  # compact wire: per chunk, the host arrays its records are gathered from

        self.cfg = cfg
        # a bare "cuda" means the process's CURRENT device, once and for all: the native publisher thread binds to an index
        # (hipSetDevice) and must wait on / copy from the device the loop's kernels run on, not device 0
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.stager = self.down_stager = None
        if batched_transfers and torch.device(device).type == "cuda":
            from .staging import PinnedStager

            self.stager = PinnedStager(device, slots=4)
            # downloads (packed blocks on their way to the log) never share a slot with uploads: a host view of a published
            # drain stays valid however many chunks are ingested meanwhile
            self.down_stager = PinnedStager(device, slots=2)
        if self.wire == "compact" and self.down_stager is None:
            from .staging import PinnedStager

            self.down_stager = PinnedStager(device, slots=2)  # K5's per-sequence outputs come back through it
        self.prof: dict[str, float] | None = {} if profile else None
        self._kernel_events: list = []
        self.overlap_publish = bool(overlap_publish) and self.stager is not None
        self._pub = None                      # prl_publisher handle while run() is active and the writer allows it
        self._pub_logs: list = []             # per partition: the prl_log handle the publisher appends to
        self._pub_inflight: deque = deque()   # (ticket, what must stay alive until that job is in the logs)
        self.publisher_ns = (0, 0)            # worker time of the last run: (whole jobs, their device -> host copies)
        # profiling only: (seconds since run() started, raw chunks waiting, samples in the ring, samples published but not yet trained on)
        self.gauges: list[tuple[float, int, int, int]] = []
        self.backpressure_waits = 0
        self._t_run = self._t_gauge = 0.0
        self.trainer_state = trainer_state
        self.ref_model = ref_model
        self.oov_patcher = oov_patcher
        published = 0
        if trainer_state is not None and getattr(trainer_state, "samples_processed", None) is not None:
            published = int(trainer_state.samples_processed)  # resume where the trainer is (:458)
        self.sched = MicroBatchScheduler(cfg.num_trainers, cfg.train_batch_size, cfg.gradient_accumulation_passes,
                                         cfg.seq_length, seq_parallel=cfg.seq_parallel, seq_packing=cfg.seq_packing,
                                         published_samples=published, length_of=lambda s: s.length)
        self.aggregator = SlidingWindowAggregator(window_size=max(10, 1000 // cfg.chunk_n_groups))
        self.ring = ProcessedRing(cfg.ring_buffer_size, cfg.drops_old_data, self.aggregator)
        self.sched.queue = self.ring.entries  # the scheduler consumes the head of the ring
        self.buffer: deque = deque()
        self.in_spec = SingleStreamSpec(exp_path=cfg.exp_path, topic=cfg.input_topic)
        self.out_spec = StreamRangeSpec(exp_path=cfg.exp_path, topic=cfg.output_topic, partition_range=(0, max(cfg.num_trainers, 1)))
        self.stats_spec = SingleStreamSpec(exp_path=cfg.exp_path, topic=cfg.stats_topic)
        self.chunks: dict[int, PreparedRollouts] = {}
        self._next_chunk = 0
        self.filtered_out = 0
        self.total_filtered_out = 0
        self.last_published_samples = published
        self.loader: ChunkLoader | None = None

    @property
    def max_model_version(self) -> int:
        return self.ring.max_model_version or 0

    def _kernels(self, name: str):
        """Profiling only: a HIP event pair around the launches of one kernel family (K5 / K6), on the current stream."""
        import contextlib

        if self.prof is None or torch.device(self.device).type != "cuda":
            return contextlib.nullcontext()
        loop = self

        class _Ctx:
            def __enter__(self):
                self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.a.record()

            def __exit__(self, *exc):
                self.b.record()
                loop._kernel_events.append((name, self.a, self.b))

        return _Ctx()

    def kernel_seconds(self) -> dict[str, float]:
        """Device time per kernel family from the event pairs collected while profiling (synchronises)."""
        if torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)
        out: dict[str, float] = {}
        for name, a, b in self._kernel_events:
            out[name] = out.get(name, 0.0) + a.elapsed_time(b) * 1e-3
        return out

    def _gauge(self, raw_chunks: int) -> None:
        """Queue depths of the loop, at most one sample per 20 ms (profiling only)."""
        now = time.perf_counter()
        if self.prof is None or now - self._t_gauge < 0.02:
            return
        self._t_gauge = now
        ts = self.trainer_state
        done = ts.samples_processed if ts is not None and ts.samples_processed is not None else 0
        self.gauges.append((now - self._t_run, raw_chunks, len(self.ring.entries) + len(self.buffer), self.sched.published_samples - int(done)))

    def _tick(self, name: str, t0: float) -> float:
        """Charge the time since `t0` to phase `name` (profiling only); returns now."""
        now = time.perf_counter()
        if self.prof is not None:
            self.prof[name] = self.prof.get(name, 0.0) + (now - t0)
        return now

    def _ingest(self, groups: list) -> None:
        """`groups`: stream records, each either a list of TrainingText dicts (the reference's text
        record) or a `RaggedRollouts` (binary record of the shm backend)."""
        from .finetune.rl import plan_groups

        t = time.perf_counter()
        if all(isinstance(g, RaggedRollouts) for g in groups):
            host = concat_ragged(groups)
        else:
            host = RaggedRollouts.from_entries([e for g in groups for e in g])
        t = self._tick("ingest_flatten", t)
        if self.stager is not None and not host.tokens.is_cuda:
            plan = plan_groups(host.host_group_index, host.host_step_index, host.host_rollout_index)
            t = self._tick("k5_plan", t)
            rag, plan_dev = host.to(self.device, stager=self.stager, extra=plan)
            t = self._tick("h2d", t)
        else:
            rag, plan_dev = host.to(self.device), None
            t = self._tick("h2d", t)
        if self.oov_patcher is not None:
            self.oov_patcher.apply(rag)
            if self.wire == "compact" and not host.tokens.is_cuda:
                # the compact records are gathered from the HOST copy of the chunk: bring the patched ids back (4 bytes per token of
                # this chunk, one copy) so that both wires publish the same tokens
                host.tokens.copy_(self.down_stager.download(rag.tokens))
                t = self._tick("oov_d2h", t)
        with self._kernels("K5"):
            prep = populate_rl_data_ragged(rag, self.cfg.eos_token_id, self.cfg.rl, plan=plan_dev)
        t = self._tick("k5_launch", t)
        keep = np.ones(prep.rollouts.n_seqs, dtype=bool)
        self.filtered_out = 0
        if self.cfg.rl.filter_zero_advantage_groups:
            keep = nonzero_advantage_mask(prep)
            self.filtered_out = int((~keep).sum())
            self.total_filtered_out += self.filtered_out
        cid = self._next_chunk
        self._next_chunk += 1
        self.chunks[cid] = prep
        if self.wire == "compact":
            if host.tokens.is_cuda:
                raise RuntimeError("compact wire: the chunk's rollouts must arrive as host records")
            self.host_chunks[cid] = {"rollouts": host, "scalars": None}
        lens = prep.rollouts.seq_lengths()
        versions = prep.rollouts.host_model_version
        self.buffer.extend(_Sample(cid, i, int(lens[i]), int(versions[i])) for i in range(len(lens)) if keep[i])
        self._tick("schedule", t)

    def _publish(self, writer) -> bool:
        """Drain the scheduler once and write what it emitted.  Returns batch_done."""
        from .finetune.data import pack_prepared

        t = time.perf_counter()
        sp = self.cfg.seq_parallel
        mbs, done = self.sched.drain()
        real = [mb for mb in mbs if not mb.sentinel]
        t = self._tick("schedule", t)
        packed: Any = None
        merged = base = None
        if self.wire == "compact":
            ref_cols = None
            if self.ref_model is not None and real:
                # KL on: the reference policy needs the packed micro-batches on this GPU (K6 + its forward, as on the full wire); only
                # the column it writes goes back to the host
                from .finetune.rl import annotate_ref_logprobs

                used = sorted({s.chunk for mb in real for s in mb.samples})
                merged = concat_prepared([self.chunks[c] for c in used])
                base, acc = {}, 0
                for c in used:
                    base[c] = acc
                    acc += self.chunks[c].rollouts.n_seqs
                pads = [(-sum(s.length for s in mb.samples)) % sp for mb in real] if sp > 1 else None
                with self._kernels("K6"):
                    packed = pack_prepared(merged, [[base[s.chunk] + s.index for s in mb.samples] for mb in real], self.cfg.eos_token_id,
                                           sentinel_pad=pads, stager=self.stager)
                t = self._tick("k6_plan_launch", t)
                for k in range(len(packed)):
                    b = packed[k]
                    annotate_ref_logprobs(self.ref_model, b, self.cfg.rl.temperature)
                    t0_, t1_ = int(packed.token_off[k]), int(packed.token_off[k + 1])
                    packed.flat["ref_logprobs"][t0_:t1_].copy_(b.ref_logprobs.reshape(-1))
                ref_cols = (packed.flat["ref_logprobs"], [int(x) for x in packed.token_off], packed)
                t = self._tick("ref_logprobs", t)
            self._submit_compact(mbs, ref_cols)
            t = self._tick("publish_submit", t)
            self._prune_chunks()
            self._tick("schedule", t)
            return done
        if real:
            used = sorted({s.chunk for mb in real for s in mb.samples})
            merged = concat_prepared([self.chunks[c] for c in used])
            base, acc = {}, 0
            for c in used:
                base[c] = acc
                acc += self.chunks[c].rollouts.n_seqs
            if self.cfg.seq_packing:
                # sequence-parallel filler: the packed length must divide by seq_parallel (data.py:222-230)
                pads = [(-sum(s.length for s in mb.samples)) % sp for mb in real] if sp > 1 else None
                with self._kernels("K6"):  # incl. the plan upload
                    packed = pack_prepared(merged, [[base[s.chunk] + s.index for s in mb.samples] for mb in real], self.cfg.eos_token_id,
                                           sentinel_pad=pads, stager=self.stager)
                t = self._tick("k6_plan_launch", t)
                if self.ref_model is not None:
                    from .finetune.rl import annotate_ref_logprobs

                    for k in range(len(packed)):  # on the device, written back into the packed block
                        b = packed[k]
                        annotate_ref_logprobs(self.ref_model, b, self.cfg.rl.temperature)
                        t0_, t1_ = int(packed.token_off[k]), int(packed.token_off[k + 1])
                        packed.flat["ref_logprobs"][t0_:t1_].copy_(b.ref_logprobs.reshape(-1))
                        # the cached batch object must keep VIEWING the block: its fields are located inside it when the record is described
                        # (sequence-parallel slices go through `describe_batch`, which refuses a tensor that lives elsewhere)
                        b.ref_logprobs = packed.flat["ref_logprobs"][t0_:t1_].reshape(1, -1)
                    t = self._tick("ref_logprobs", t)
        job = {"mbs": mbs, "packed": packed, "merged": merged, "base": base, "max_model_version": self.max_model_version}
        if self._pub is not None:
            self._submit(job)
            t = self._tick("publish_submit", t)
        else:
            self._write_out(writer, job, self.down_stager)
            t = time.perf_counter()
        self._prune_chunks()
        self._tick("schedule", t)
        return done

    def _batches_of(self, job: dict, packed):
        """(trainer partition, batch) of a drain in the scheduler's order, sequence-parallel slices side by side."""
        from .finetune.data import pad_prepared
        from .finetune.utils import create_sentinel_batch

        sp = self.cfg.seq_parallel
        merged, base = job["merged"], job["base"]
        k = 0
        for mb in job["mbs"]:
            if mb.sentinel:
                batch = create_sentinel_batch(None, tokenizer=type("T", (), {"eos_token_id": self.cfg.eos_token_id})(), model_version=job["max_model_version"])
            elif self.cfg.seq_packing:
                batch = packed[k]
                k += 1
            else:  # unpacked: fixed train_batch_size rows padded to a common length (reference `collate`, :639-648)
                batch = pad_prepared(merged, [base[s.chunk] + s.index for s in mb.samples], padding_side=self.cfg.padding_side)
                if self.ref_model is not None:
                    from .finetune.rl import annotate_ref_logprobs

                    annotate_ref_logprobs(self.ref_model, batch, self.cfg.rl.temperature)
            slices = batch.make_slices(sp) if sp > 1 else [batch]
            for off, piece in enumerate(slices):
                yield mb.trainer_id + off, piece

    def _write_out(self, writer, job: dict, stager) -> None:
        """Inline second half of a drain: the packed block leaves the device in ONE copy, every micro-batch (and sentinel) is
        framed and appended to its trainer's partition, in the scheduler's order."""
        packed = job["packed"]
        t = time.perf_counter()
        if packed is not None and stager is not None and packed.block is not None:
            packed = packed.to_host(stager)  # ONE device -> host copy for every micro-batch of this drain
            t = self._tick("d2h", t)
        for partition, piece in self._batches_of(job, packed):
            writer.write(piece, partition=partition)
        self._tick("encode_publish", t)

    def _submit(self, job: dict) -> None:
        """Hand a drain to the native publisher: record headers and the piece table are built here (host arithmetic only), the
        device -> host copy and the gathering into the logs happen on its thread."""
        import ctypes

        from . import _lib, batch_codec

        lib = _lib.load()
        packed = job["packed"]
        block = packed.block if packed is not None else None
        block_ptr, block_nbytes = (block.data_ptr(), block.numel()) if block is not None else (0, 0)
        inline = bytearray()
        recs, pieces = [], []
        if self.cfg.seq_parallel == 1 and block is not None:
            # the common case by arithmetic on the block's geometry: no per-micro-batch tensor views
            from .finetune.utils import create_sentinel_batch

            k = 0
            for mb in job["mbs"]:
                if mb.sentinel:
                    s_batch = create_sentinel_batch(None, tokenizer=type("T", (), {"eos_token_id": self.cfg.eos_token_id})(), model_version=job["max_model_version"])
                    nbytes, ps = batch_codec.describe_batch(s_batch, 0, 0, inline)
                else:
                    nbytes, ps = packed.describe_record(k, inline)
                    k += 1
                recs.append((self._pub_logs[mb.trainer_id], nbytes, len(pieces), len(ps)))
                pieces += ps
        else:
            for partition, piece in self._batches_of(job, packed):
                nbytes, ps = batch_codec.describe_batch(piece, block_ptr, block_nbytes, inline)
                recs.append((self._pub_logs[partition], nbytes, len(pieces), len(ps)))
                pieces += ps
        # the two tables as numpy records laid out like `prl_pub_record` / `prl_pub_piece` (building ~230 ctypes structs costs 3 x as much)
        rec_arr = np.array(recs, dtype=_PUB_RECORD_DT)
        piece_arr = np.array([(src, off, nb, kind, 0) for kind, src, off, nb in pieces], dtype=_PUB_PIECE_DT)
        ready = None
        if block is not None:
            with torch.cuda.device(self.device):  # the event belongs to the loop's device, whatever the caller's current device is
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(self.device))  # K6 (and the reference-policy annotation) of this drain are complete once it has fired
        inline_c = (ctypes.c_char * len(inline)).from_buffer(inline) if inline else None
        ticket = ctypes.c_uint64()
        _lib.check(lib.prl_publisher_submit(self._pub, block_ptr or None, block_nbytes, ready.cuda_event if ready is not None else None,
                                            rec_arr.ctypes.data, len(recs), piece_arr.ctypes.data, len(pieces), inline_c, len(inline), ctypes.byref(ticket)))
        self._pub_inflight.append((ticket.value, block, ready))
        done = ctypes.c_uint64()
        _lib.check(lib.prl_publisher_completed(self._pub, ctypes.byref(done)))
        while self._pub_inflight and self._pub_inflight[0][0] <= done.value:
            self._pub_inflight.popleft()  # its block goes back to the allocator

    def _chunk_sources(self, cid: int) -> dict:
        """Compact wire: a chunk's gather sources, built the first time a micro-batch needs the chunk - K5's four outputs come back
        in ONE small device -> host copy (16 bytes per sequence)."""
        hc = self.host_chunks[cid]
        if hc["scalars"] is None:
            k5 = self.down_stager.download(self.chunks[cid].k5_out32).numpy()  # rows: num_labels, overflow, advantage, group_tokens
            hc.update(compact_sources(hc["rollouts"], k5))
        return hc

    def _submit_compact(self, mbs: list, ref_cols: tuple | None = None) -> None:
        """One drain on the compact wire: per micro-batch a PRLCMP01 record whose per-token columns the publisher copies from
        the chunks' host arrays (PRL_PUB_FROM_HOST pieces, one per sequence and column), header and per-sequence arrays inline.
        Sequence parallelism: the record goes to every partition of the lead trainer's SP group, each copy naming its slice.
        `ref_cols` = (flat fp32 device column, token offsets of the drain's real micro-batches, owner): the job's device block; every
        record takes its micro-batch's range of it as `ref_column`."""
        import ctypes

        from . import _lib, batch_codec
        from .finetune.utils import create_sentinel_batch

        if not mbs:
            return
        lib = _lib.load()
        sp = self.cfg.seq_parallel
        inline = bytearray()
        recs, pieces, keep = [], [], []
        k = 0
        for mb in mbs:
            if mb.sentinel:
                s_batch = create_sentinel_batch(None, tokenizer=type("T", (), {"eos_token_id": self.cfg.eos_token_id})(), model_version=self.max_model_version)
                for off, piece in enumerate(s_batch.make_slices(sp) if sp > 1 else [s_batch]):
                    first = len(pieces)
                    nbytes, ps = batch_codec.describe_batch(piece, 0, 0, inline)
                    pieces += [(src, o, nb, kind, 0) for kind, src, o, nb in ps]
                    recs.append((self._pub_logs[mb.trainer_id + off], nbytes, first, len(pieces) - first))
                continue
            members = [(self._chunk_sources(s.chunk), s.index) for s in mb.samples]
            keep += [hc["rollouts"] for hc, _ in members]
            n = sum(s.length for s in mb.samples)
            pad = (-n) % sp if sp > 1 else 0
            ref_block = None
            if ref_cols is not None:
                t0_, t1_ = ref_cols[1][k], ref_cols[1][k + 1]
                assert t1_ - t0_ == n + pad, "the packed micro-batch and its compact record disagree on the token count"
                ref_block = (4 * t0_, 4 * (t1_ - t0_))
            k += 1
            version = min(s.model_version for s in mb.samples)
            for off in range(sp):
                first = len(pieces)
                nbytes = describe_compact(members, version, self.cfg.eos_token_id, inline, pieces, padding=pad, slice_index=off, num_slices=sp, ref_block=ref_block)
                recs.append((self._pub_logs[mb.trainer_id + off], nbytes, first, len(pieces) - first))
        rec_arr = np.array(recs, dtype=_PUB_RECORD_DT)
        piece_arr = np.array(pieces, dtype=_PUB_PIECE_DT)
        inline_c = (ctypes.c_char * len(inline)).from_buffer(inline) if inline else None
        block_ptr, block_nbytes, ready = None, 0, None
        if ref_cols is not None:
            col = ref_cols[0]
            block_ptr, block_nbytes = col.data_ptr(), col.numel() * col.element_size()
            with torch.cuda.device(self.device):
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(self.device))  # K6 and the reference policy's forward of this drain are complete once it has fired
            keep.append(ref_cols)
        ticket = ctypes.c_uint64()
        _lib.check(lib.prl_publisher_submit(self._pub, block_ptr, block_nbytes, ready.cuda_event if ready is not None else None,
                                            rec_arr.ctypes.data, len(recs), piece_arr.ctypes.data, len(pieces),
                                            inline_c, len(inline), ctypes.byref(ticket)))
        self._pub_inflight.append((ticket.value, keep, ready))  # host arrays and the device column stay alive until the publisher has copied them
        done = ctypes.c_uint64()
        _lib.check(lib.prl_publisher_completed(self._pub, ctypes.byref(done)))
        while self._pub_inflight and self._pub_inflight[0][0] <= done.value:
            self._pub_inflight.popleft()

    def _start_publisher(self, writer) -> None:
        """The native publisher appends through the partition writers' own log handles; anything it cannot serve - another
        backend, a JSONL mirror that wants every record as text too - keeps the inline path."""
        import ctypes

        from . import _lib

        parts = getattr(writer, "_writers", None)
        if not parts or any(getattr(w, "_mirror", None) is not None or getattr(w, "_log", None) is None for w in parts):
            return
        h = ctypes.c_void_p()
        _lib.check(_lib.load().prl_publisher_create(int(self.device.index), ctypes.byref(h)))
        self._pub, self._pub_logs = h, [getattr(w._log._h, "value", w._log._h) for w in parts]

    def _stop_publisher(self) -> None:
        """Everything handed to the publisher is in the logs when this returns; its error, if any, is raised here."""
        import ctypes

        from . import _lib

        if self._pub is None:
            return
        lib, pub = _lib.load(), self._pub
        self._pub = None
        try:
            if self._pub_inflight:
                _lib.check(lib.prl_publisher_wait(pub, self._pub_inflight[-1][0], -1))
        finally:
            busy, copy = ctypes.c_uint64(), ctypes.c_uint64()
            lib.prl_publisher_stats(pub, ctypes.byref(busy), ctypes.byref(copy))
            self.publisher_ns = (busy.value, copy.value)
            lib.prl_publisher_destroy(pub)
            self._pub_inflight.clear()
            self._pub_logs = []

    def _prune_chunks(self) -> None:
        """Release the device-resident chunks whose samples have all been scheduled or dropped.  Called after every
        publish AND after every ring admission: while the back-pressure rule holds publishing back, the ring keeps
        dropping old samples, and their chunk tensors must not stay alive until the trainer catches up."""
        alive = {s.chunk for s in self.ring.entries} | {s.chunk for s in self.sched._current} | {s.chunk for s in self.buffer}
        for c in [c for c in self.chunks if c not in alive]:
            del self.chunks[c]
            self.host_chunks.pop(c, None)

    def _maybe_write_stats(self, stats_writer, batch_done: bool, raw_queue_chunks: int) -> None:
        pub = self.sched.published_samples
        if not should_write_stats(pub, self.last_published_samples, self.cfg.debug_mode, batch_done, self.cfg.log_every_n_samples):
            return
        # there is no worker output queue here (K5 runs inline): its two gauges read 0
        stats_writer.write(preprocessor_stats_record(pub, self.ring.max_model_version, raw_queue_chunks, 0, self.cfg.chunk_n_groups,
                                                     self.cfg.attempts, self.filtered_out, self.total_filtered_out, self.aggregator))
        self.last_published_samples = pub
        self.filtered_out = 0

    def run(self, max_published_samples: int | None = None, idle_timeout: float = 5.0) -> int:
        """Consume the actor stream until the trainer reports `samples_target` processed samples (the
        reference's stop rule, :513-518), `max_published_samples` were published, or the stream stays
        silent for `idle_timeout` seconds.  Returns the number of published samples."""
        import queue
        import threading

        from .streams import write_to_streams

        cfg = self.cfg
        raw_q: queue.Queue = queue.Queue(cfg.raw_queue_size)
        self.loader = ChunkLoader(raw_q, self.in_spec, cfg.attempts, cfg.chunk_n_groups, cfg.drops_old_data)
        threading.Thread(target=self.loader.run, name="preprocessor-loader", daemon=True).start()
        start = self.sched.published_samples
        self._t_run = time.perf_counter()
        with write_to_streams(self.out_spec) as writer, write_to_streams(self.stats_spec) as stats_writer:
            if (self.overlap_publish and cfg.seq_packing) or self.wire == "compact":
                self._start_publisher(writer)
                if self.wire == "compact" and self._pub is None:
                    raise ValueError("the compact wire is written by the native publisher: it needs `backend: shm` partitions without a JSONL mirror")
            try:
                self._run_loop(raw_q, writer, stats_writer, start, max_published_samples, idle_timeout)
            finally:
                self._stop_publisher()  # every drain handed over is in the log before the writers close
        return self.sched.published_samples - start

    def _run_loop(self, raw_q, writer, stats_writer, start: int, max_published_samples: int | None, idle_timeout: float) -> None:
        import queue

        cfg, ts = self.cfg, self.trainer_state
        last_data = time.time()
        while max_published_samples is None or self.sched.published_samples - start < max_published_samples:
            if cfg.samples_target is not None and ts is not None and ts.samples_processed is not None and ts.samples_processed >= cfg.samples_target:
                logger.info("Trainer signalled completion; stopping preprocessor loop")
                break
            self._gauge(raw_q.qsize())
            t = time.perf_counter()
            try:
                chunk = raw_q.get(timeout=0.01)
                t = self._tick("input_wait", t)
                if isinstance(chunk, Exception):
                    raise chunk
                self._ingest(chunk)
                last_data = time.time()
            except queue.Empty:
                self._tick("input_wait", t)
                if time.time() - last_data > idle_timeout and not self.ring.entries and not self.buffer:
                    break
            if len(self.buffer) < cfg.dataset_buffer_size:
                continue
            t = time.perf_counter()
            self.ring.admit(self.buffer)
            self._prune_chunks()
            self._tick("schedule", t)
            if ts is not None and ts.samples_processed is not None:
                if self.sched.published_samples - ts.samples_processed > cfg.max_ready_samples_per_lead * cfg.num_trainers:
                    self.backpressure_waits += 1
                    continue  # wait for the finetune loop to catch up
            batch_done = False
            while self.ring.entries and not batch_done:
                before = self.sched.published_samples
                batch_done = self._publish(writer)
                if self.sched.published_samples == before and not batch_done:
                    break
                if max_published_samples is not None and self.sched.published_samples - start >= max_published_samples:
                    break
            self._maybe_write_stats(stats_writer, batch_done, raw_q.qsize())
