"""Return types of the rollout plugin surface (reference pipelinerl/rollouts.py:6-97).

A rollout policy is `async def f(cfg, llm, problem, session) -> RolloutResult` resolved from
`cfg.actor.rollout_policy`; a dataset loader is `load_problems(dataset_names, **params) ->
list[dict]` resolved from `cfg.dataset_loader` (reference actor.py:141, 803-808).  The field names
below are the wire contract: `TrainingText.model_dump()` is the `actor` stream record
 that `RaggedRollouts.from_entries` ingests.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
from pydantic import BaseModel, Field


class BaseMetrics(BaseModel):
    reward: float
    success: bool
    no_error: bool
    no_answer: bool


class TrainingText(BaseModel):
    """One (prompt, completion) training sample produced by a rollout.

    input_ids = prompt + completion token ids; labels = -100 on the prompt and the token id on the
    completion; logprobs / ref_logprobs cover completion tokens only; group_id ties the `attempts`
    rollouts of one problem together for the group baseline; metadata carries model_version,
    rollout_index and step_index (stamped by the actor, reference actor.py:210-219)."""

    model_config = {"arbitrary_types_allowed": True}

    text: str
    n_predicted: int
    reward: float = 0.0
    logprobs: List[float] = Field(default_factory=list)
    ref_logprobs: List[float] = Field(default_factory=list)
    input_ids: List[int] = Field(default_factory=list)
    labels: List[int] = Field(default_factory=list)
    group_id: Optional[str] = None
    finished: bool = False
    prompt_tokens: int = 0
    output_tokens: int = 0
    visual_features: Optional[Dict[str, np.ndarray]] = None
    metadata: dict = Field(default_factory=dict)

    @property
    def prompt_text(self) -> str:
        return self.text[: -self.n_predicted]

    @property
    def output_text(self) -> str:
        return self.text[-self.n_predicted :]

    def check_consistency(self) -> None:
        """What the preprocess path relies on (rl/__init__.py:582-585, preprocess.py:90-92):
        one label per token, one logprob per target token, targets form the tail of the sequence."""
        if len(self.labels) != len(self.input_ids):
            raise ValueError(f"labels ({len(self.labels)}) and input_ids ({len(self.input_ids)}) differ in length")
        n_targets = sum(1 for x in self.labels if x != -100)
        if n_targets != len(self.logprobs):
            raise ValueError(f"Target tokens: {n_targets}, old logprobs: {len(self.logprobs)}")
        if self.ref_logprobs and len(self.ref_logprobs) != len(self.logprobs):
            raise ValueError(f"{len(self.ref_logprobs)} != {len(self.logprobs)}")


class RolloutResult(BaseModel):
    training_texts: list[TrainingText]
    metrics: BaseMetrics
    latency: float
    # filled in by the actor after the policy returns
    model_version: Optional[int] = None
    dataset_name: Optional[str] = None
    group_id: Optional[str] = None
    domain: Optional[str] = None


def stamp_group(results: Sequence["RolloutResult"], group_id: str, model_version: int) -> list[dict]:
    """What the actor does to a finished group before publishing it (reference actor.py:210-225,
    648-652): every training text gets `group_id` and metadata {model_version, rollout_index,
    step_index}; the returned list of plain dicts is ONE record of the `actor` stream."""
    record: list[dict] = []
    for rollout_index, result in enumerate(results):
        result.model_version = model_version
        result.group_id = group_id
        for step_index, text in enumerate(result.training_texts):
            text.group_id = group_id
            text.metadata.update(model_version=model_version, rollout_index=rollout_index, step_index=step_index)
            record.append(text.model_dump())
    return record


@dataclass(frozen=True)
class TrainingTextSummary:
    prompt_tokens: list[int]
    output_tokens: list[int]
    overflow: bool
    num_turns: int


def apply_rollout_reward(training_texts: Sequence[TrainingText], reward: float) -> list[TrainingText]:
    out = list(training_texts)
    for t in out:
        t.reward = reward
    return out


def rollout_has_overflow(training_texts: Sequence[TrainingText]) -> bool:
    return not all(t.finished for t in training_texts)


def summarize_training_texts(training_texts: Sequence[TrainingText]) -> TrainingTextSummary:
    ts = list(training_texts)
    return TrainingTextSummary(
        prompt_tokens=[t.prompt_tokens for t in ts],
        output_tokens=[t.output_tokens for t in ts],
        overflow=rollout_has_overflow(ts),
        num_turns=len(ts),
    )


def resolve_plugin(dotted_path: str):
    """`pkg.module.function` -> the callable (stand-in for hydra.utils.get_method, actor.py:141)."""
    import importlib

    module, _, attr = dotted_path.rpartition(".")
    if not module:
        raise ValueError(f"not a dotted path: {dotted_path!r}")
    return getattr(importlib.import_module(module), attr)
