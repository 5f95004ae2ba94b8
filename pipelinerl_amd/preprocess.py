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

from .finetune.rl import PreparedRollouts, RLConfig, populate_rl_data_ragged
from .ragged import RaggedRollouts


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
