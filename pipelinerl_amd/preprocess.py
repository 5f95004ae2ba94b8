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
        self.queue.extend(samples)

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


@dataclass
class PreprocessorConfig:
    """The keys of the reference config the preprocessor loop consumes (conf/base.yaml:25-44,
    conf/finetune/base.yaml; SURVEY.md §5 "config")."""

    exp_path: Any
    num_trainers: int
    train_batch_size: int
    gradient_accumulation_passes: int
    seq_length: int
    attempts: int
    rl: RLConfig
    eos_token_id: int
    seq_parallel: int = 1
    chunk_n_groups: int = 2
    max_ready_samples_per_lead: int = 64
    input_topic: str = "actor"
    output_topic: str = "training_data"


@dataclass
class _Sample:
    chunk: int
    index: int
    length: int


class PreprocessorLoop:
    """`run_preprocessing_loop` of the reference (preprocess.py:370-704) without worker processes:
    the per-chunk work that needed N CPU workers is two kernel launches here.

        actor stream -> chunks of `chunk_n_groups` groups -> K5 on device -> MicroBatchScheduler
        -> ONE K6 launch per drain -> per-trainer `training_data` partitions (+ sentinels)

    Back-pressure as in the reference (:587-592): publishing pauses while
    published - trainer_state.samples_processed exceeds max_ready_samples_per_lead * num_trainers."""

    def __init__(self, cfg: PreprocessorConfig, device, trainer_state=None, ref_model=None):
        """`ref_model`: a frozen reference policy on the preprocessor's GPU.  When given, every real
        micro-batch gets its `ref_logprobs` from a no-grad forward of that model (K1 on its logits)
        before it is published - the device-side replacement of the reference's HTTP round trip to a
        second inference server (preprocess.py:86-104, llm.py:606-648)."""
        from .streams import SingleStreamSpec, StreamRangeSpec

        self.cfg = cfg
        self.device = device
        self.trainer_state = trainer_state
        self.ref_model = ref_model
        self.sched = MicroBatchScheduler(cfg.num_trainers, cfg.train_batch_size, cfg.gradient_accumulation_passes,
                                         cfg.seq_length, seq_parallel=cfg.seq_parallel, length_of=lambda s: s.length)
        self.in_spec = SingleStreamSpec(exp_path=cfg.exp_path, topic=cfg.input_topic)
        self.out_spec = StreamRangeSpec(exp_path=cfg.exp_path, topic=cfg.output_topic, partition_range=(0, max(cfg.num_trainers, 1)))
        self.chunks: dict[int, PreparedRollouts] = {}
        self._next_chunk = 0
        self.filtered_out = 0
        self.max_model_version = 0

    def _ingest(self, groups: list) -> None:
        """`groups`: stream records, each either a list of TrainingText dicts (the reference's text
        record) or a `RaggedRollouts` (binary record of the shm backend)."""
        if all(isinstance(g, RaggedRollouts) for g in groups):
            rag = concat_ragged(groups).to(self.device)
            sizes = np.bincount(rag.host_group_index)
            pairs = np.unique(np.stack([rag.host_group_index, rag.host_rollout_index]), axis=1)
            if not (np.bincount(pairs[0], minlength=len(sizes)) == self.cfg.attempts).all():
                raise ValueError("Group sizes are wrong")
            prep = populate_rl_data_ragged(rag, self.cfg.eos_token_id, self.cfg.rl)
        else:
            entries = [e for g in groups for e in g]
            if not check_group_sizes(entries, self.cfg.attempts):
                raise ValueError("Group sizes are wrong")
            prep = preprocess_chunk(entries, self.cfg.eos_token_id, self.cfg.rl, self.device)
        keep = np.ones(prep.rollouts.n_seqs, dtype=bool)
        if self.cfg.rl.filter_zero_advantage_groups:
            keep = nonzero_advantage_mask(prep)
            self.filtered_out += int((~keep).sum())
        cid = self._next_chunk
        self._next_chunk += 1
        self.chunks[cid] = prep
        lens = prep.rollouts.seq_lengths()
        self.max_model_version = max(self.max_model_version, int(prep.rollouts.host_model_version.max()))
        self.sched.push(_Sample(cid, i, int(lens[i])) for i in range(len(lens)) if keep[i])

    def _publish(self, writer) -> bool:
        """Drain the scheduler once and write what it emitted.  Returns batch_done."""
        from .finetune.data import pack_prepared
        from .finetune.utils import create_sentinel_batch

        mbs, done = self.sched.drain()
        real = [mb for mb in mbs if not mb.sentinel]
        packed = None
        if real:
            used = sorted({s.chunk for mb in real for s in mb.samples})
            merged = concat_prepared([self.chunks[c] for c in used])
            base, acc = {}, 0
            for c in used:
                base[c] = acc
                acc += self.chunks[c].rollouts.n_seqs
            packed = pack_prepared(merged, [[base[s.chunk] + s.index for s in mb.samples] for mb in real], self.cfg.eos_token_id)
        k = 0
        for mb in mbs:
            if mb.sentinel:
                batch = create_sentinel_batch(None, tokenizer=type("T", (), {"eos_token_id": self.cfg.eos_token_id})(), model_version=self.max_model_version)
            else:
                batch = packed[k]
                k += 1
                if self.ref_model is not None:
                    from .finetune.rl import annotate_ref_logprobs

                    annotate_ref_logprobs(self.ref_model, batch, self.cfg.rl.temperature)
            slices = batch.make_slices(self.cfg.seq_parallel) if self.cfg.seq_parallel > 1 else [batch]
            for off, piece in enumerate(slices):
                writer.write(piece, partition=mb.trainer_id + off)
        # chunks whose samples have all been scheduled can be dropped
        alive = {s.chunk for s in self.sched.queue} | {s.chunk for s in self.sched._current}
        for c in [c for c in self.chunks if c not in alive]:
            del self.chunks[c]
        return done

    def run(self, max_published_samples: int | None = None, idle_timeout: float = 5.0) -> int:
        """Consume the actor stream until `max_published_samples` were published (or the stream stays
        silent for `idle_timeout` seconds).  Returns the number of published samples."""
        import queue
        import threading

        from .streams import read_stream, write_to_streams

        groups_q: queue.Queue = queue.Queue()

        def reader():
            with read_stream(self.in_spec) as r:
                for record in r.read():
                    groups_q.put(record)

        threading.Thread(target=reader, daemon=True).start()
        pending: list[list[dict]] = []
        start = self.sched.published_samples
        last_data = time.time()
        with write_to_streams(self.out_spec) as writer:
            while max_published_samples is None or self.sched.published_samples - start < max_published_samples:
                try:
                    pending.append(groups_q.get(timeout=0.01))
                    last_data = time.time()
                except queue.Empty:
                    if time.time() - last_data > idle_timeout and not self.sched.queue:
                        break
                if len(pending) >= self.cfg.chunk_n_groups:
                    self._ingest(pending)
                    pending = []
                ts = self.trainer_state
                if ts is not None and ts.samples_processed is not None:
                    limit = self.cfg.max_ready_samples_per_lead * self.cfg.num_trainers
                    if self.sched.published_samples - ts.samples_processed > limit:
                        continue
                while self.sched.queue:
                    before = self.sched.published_samples
                    done = self._publish(writer)
                    if self.sched.published_samples == before and not done:
                        break
                    if max_published_samples is not None and self.sched.published_samples - start >= max_published_samples:
                        break
        return self.sched.published_samples - start
