"""The caller of the rollout / dataset plugin surface (reference pipelinerl/actor.py:141, 176-225, 648-652,
803-808) - NOT an actor.

The reference's actor process is an I/O-bound asyncio program around vLLM's HTTP API (SURVEY.md §2 rows 15-16:
out of scope).  What the hot path needs from it is the CONTRACT at its two plugin seams and at its output:

  * `cfg.dataset_loader`  -> `load_problems(dataset_names, **cfg.dataset_loader_params) -> list[dict]`   (actor.py:803-808)
  * `cfg.actor.rollout_policy` -> `async policy(cfg, llm, problem, session) -> RolloutResult`, resolved by dotted path
    (actor.py:141); a plain function returning the result is accepted too (domains/dispatcher.py:84-86)
  * every problem is rolled out `attempts` times = one GROUP; each rollout goes to the llm with the fewest rollouts in
    flight (actor.py:247-262); retryable exceptions are retried with exponential back-off (actor.py:146-153, 190-207),
    anything else stops the run
  * the finished rollout is stamped: `model_version` = the trainer's propagated weight version at the START of the
    rollout, `group_id` = "<scheduler>_<group>", per training text `metadata.{model_version, rollout_index,
    step_index}` (actor.py:210-219); a complete group is shuffled (actor.py:222) and published as ONE record of the
    `actor` stream: the list of `TrainingText.model_dump()` of all its rollouts (actor.py:648-652)

`ActorHarness` does exactly that and nothing else (no vLLM, no HTTP, no stats, no test loop, no domain sampler), so a
user's `generate_rollout` / `load_problems` pair can be driven end to end into `PreprocessorLoop` -> `LearnerStep`
without the reference's actor.  With `wire="ragged"` the group travels as the binary SoA record (`PRLROL01`,
batch_codec.py) that the shm backend mirrors as JSONL; with `wire="jsonl"` as the reference's text record.
"""

from __future__ import annotations

import asyncio
import inspect
import logging
import random
from pathlib import Path
from typing import Any, Callable, Sequence

from .rollouts import RolloutResult, resolve_plugin
from .streams import SingleStreamSpec, write_to_streams

logger = logging.getLogger(__name__)

RETRYABLE_ROLLOUT_EXCEPTIONS: tuple[type[BaseException], ...] = (asyncio.TimeoutError, TimeoutError)


def cfg_get(cfg: Any, dotted: str, default: Any = None) -> Any:
    """`cfg.a.b` for OmegaConf nodes, namespaces and plain dicts alike; `default` when a key is missing."""
    node = cfg
    for key in dotted.split("."):
        if node is None:
            return default
        if isinstance(node, dict):
            node = node.get(key, None)
        elif hasattr(node, "get") and not hasattr(node, key):
            node = node.get(key, None)
        else:
            node = getattr(node, key, None)
    return default if node is None else node


class ActorHarness:
    """Resolve the two plugins of `cfg`, roll out groups on `llms`, publish them to the `actor` stream."""

    def __init__(self, cfg: Any, llms: Sequence[Any], exp_path: str | Path, trainer_state: Any = None,
                 scheduler_name: str = "actor0", wire: str = "jsonl", session: Any = None, shuffle_seed: int | None = None,
                 retryable: tuple[type[BaseException], ...] = RETRYABLE_ROLLOUT_EXCEPTIONS):
        if wire not in ("jsonl", "ragged"):
            raise ValueError(f"wire must be 'jsonl' or 'ragged', got {wire!r}")
        if not llms:
            raise ValueError("at least one llm is needed")
        self.cfg, self.llms, self.session = cfg, list(llms), session
        self.exp_path = Path(exp_path)
        self.trainer_state = trainer_state
        self.scheduler_name, self.wire = scheduler_name, wire
        self.attempts = int(cfg_get(cfg, "attempts", 1))
        self.rollout_policy: Callable = resolve_plugin(str(cfg_get(cfg, "actor.rollout_policy")))
        self.dataset_loader: Callable = resolve_plugin(str(cfg_get(cfg, "dataset_loader")))
        self.max_rollout_retries = int(cfg_get(cfg, "actor.max_rollout_retries", -1))  # -1: retry forever
        self.retry_initial_delay_s = float(cfg_get(cfg, "actor.rollout_retry_initial_delay_s", 1.0))
        self.retry_max_delay_s = float(cfg_get(cfg, "actor.rollout_retry_max_delay_s", 30.0))
        self.retryable = retryable
        self.data_stream = SingleStreamSpec(exp_path=self.exp_path, topic="actor")
        self._active = [0] * len(self.llms)
        self._rng = random.Random(shuffle_seed) if shuffle_seed is not None else random
        self.published_samples = 0
        self.published_groups = 0
        self.retries = 0
        self.timing: dict[str, float] = {}

    # -- plugins ---------------------------------------------------------------------------------
    def load_problems(self, split: str = "train") -> list[dict]:
        """`dataset_loader(cfg.<split>_dataset_names, **cfg.dataset_loader_params)` + `cfg.train_subset` (actor.py:803-811)."""
        names = list(cfg_get(self.cfg, f"{split}_dataset_names", []))
        params = dict(cfg_get(self.cfg, "dataset_loader_params", {}) or {})
        problems = self.dataset_loader(names, **params)
        subset = cfg_get(self.cfg, "train_subset") if split == "train" else None
        if subset:
            problems = problems[cfg_get(subset, "begin"): cfg_get(subset, "end")]
        return problems

    def _model_version(self) -> int:
        if self.trainer_state is None:
            return 0
        v = self.trainer_state.propagated_weight_version
        assert v is not None, "the trainer has not announced a weight version yet"
        return int(v)

    async def _one_rollout(self, problem: dict, group_id: int, rollout_index: int) -> RolloutResult:
        llm_index = min(range(len(self.llms)), key=lambda i: self._active[i])  # the least busy llm (actor.py:247-262)
        self._active[llm_index] += 1
        try:
            model_version = self._model_version()
            retry = 0
            while True:
                try:
                    out = self.rollout_policy(self.cfg, self.llms[llm_index], problem, self.session)
                    result = await out if inspect.isawaitable(out) else out
                    break
                except asyncio.CancelledError:
                    raise
                except Exception as exc:  # noqa: BLE001 - the reference's rule: retry the retryable, stop on the rest
                    if isinstance(exc, self.retryable) and (self.max_rollout_retries < 0 or retry < self.max_rollout_retries):
                        retry += 1
                        self.retries += 1
                        await asyncio.sleep(min(self.retry_max_delay_s, self.retry_initial_delay_s * 2 ** (retry - 1)))
                        continue
                    raise
            if not isinstance(result, RolloutResult):
                result = RolloutResult.model_validate(result.model_dump() if hasattr(result, "model_dump") else result)
            full_group_id = f"{self.scheduler_name}_{group_id}"
            result.model_version = model_version
            result.group_id = full_group_id
            for step_index, sample in enumerate(result.training_texts):
                sample.metadata["model_version"] = model_version
                sample.metadata["rollout_index"] = rollout_index
                sample.metadata["step_index"] = step_index
                sample.group_id = full_group_id
            return result
        finally:
            self._active[llm_index] -= 1

    async def rollout_group(self, problem: dict, group_id: int) -> list[RolloutResult]:
        """`attempts` concurrent rollouts of one problem, stamped and shuffled like a finished group of the reference."""
        group = list(await asyncio.gather(*[self._one_rollout(problem, group_id, k) for k in range(self.attempts)]))
        self._rng.shuffle(group)
        return group

    # -- publishing ------------------------------------------------------------------------------
    def group_record(self, group: Sequence[RolloutResult]) -> list[dict]:
        """ONE `actor` stream record: every training text of every rollout of the group (actor.py:648-652)."""
        return [text.model_dump() for r in group for text in r.training_texts]

    def publish(self, writer: Any, group: Sequence[RolloutResult]) -> int:
        record = self.group_record(group)
        if self.wire == "ragged":
            from .ragged import RaggedRollouts

            writer.write(RaggedRollouts.from_entries(record))
        else:
            writer.write(record)
        self.published_samples += len(record)
        self.published_groups += 1
        return len(record)

    async def run_async(self, problems: Sequence[dict], n_groups: int | None = None, first_group_id: int = 0,
                        concurrent_groups: int = 4) -> int:
        """Roll out `n_groups` problems (default: each problem once, in order), publishing every group as it completes."""
        if not problems:
            raise ValueError("no problems to roll out (the dataset loader returned an empty list)")
        todo = list(problems if n_groups is None else [problems[i % len(problems)] for i in range(n_groups)])
        start = self.published_samples
        with write_to_streams(self.data_stream) as writer:
            for lo in range(0, len(todo), concurrent_groups):
                batch = todo[lo: lo + concurrent_groups]
                groups = await asyncio.gather(*[self.rollout_group(p, first_group_id + lo + k) for k, p in enumerate(batch)])
                for g in groups:
                    self.publish(writer, g)
        return self.published_samples - start

    def run(self, problems: Sequence[dict] | None = None, n_groups: int | None = None, **kw: Any) -> int:
        return asyncio.run(self.run_async(self.load_problems() if problems is None else problems, n_groups, **kw))

    # -- the training actor loop's pacing (actor.py:510-557) ----------------------------------------
    @staticmethod
    def submission_budget(attempts: int, train_batch_size: int, gradient_accumulation_passes: int, weight_update_interval: int,
                          max_lag: int | None) -> tuple[float, int | None]:
        """(groups that may be submitted before the first weight update, groups added per weight update) - the arithmetic of
        actor.py:510-534: with `max_lag` the actor may run `ceil(max_lag / attempts)` groups ahead of what one update consumes
        (`ceil(ceil(interval / B) * B / attempts)`, B = samples per optimizer step); without it nothing holds it back."""
        import math

        if max_lag is None:
            return math.inf, None
        total_batch_size = train_batch_size * gradient_accumulation_passes
        total_update_size = math.ceil(weight_update_interval / total_batch_size) * total_batch_size
        groups_per_update = math.ceil(total_update_size / attempts)
        return math.ceil(max_lag / attempts) + groups_per_update, groups_per_update

    async def run_paced_async(self, problems: Sequence[dict], samples_target: int, train_batch_size: int, gradient_accumulation_passes: int,
                              weight_update_interval: int = 1, max_lag: int | None = None, concurrent_groups: int = 4, poll_s: float = 0.002,
                              max_groups: int | None = None, on_group: Callable | None = None) -> int:
        """The TRAINING actor loop's rules around the same rollouts (actor.py:536-557): stop once the trainer reports
        `samples_target` processed samples; never have more than the submission budget of groups submitted (`max_lag`), the
        budget growing by one update's worth every time a new weight version has PROPAGATED (`trainer_state`); problems are
        drawn for ever (epoch after epoch - the problem dict carries `epoch`).  Needs a `trainer_state` that is being
        followed.  Fills `self.timing` (seconds rolling out + publishing, seconds held back by the lag rule)."""
        import time

        if self.trainer_state is None:
            raise ValueError("the paced loop follows the trainer: pass trainer_state")
        if not problems:
            raise ValueError("no problems to roll out (the dataset loader returned an empty list)")
        ts = self.trainer_state
        while ts.propagated_weight_version is None:  # actor.py:497 asserts it; a stage started early simply waits
            await asyncio.sleep(poll_s)
        last_version = ts.propagated_weight_version
        can_submit, per_update = self.submission_budget(self.attempts, train_batch_size, gradient_accumulation_passes, weight_update_interval, max_lag)
        submitted = 0
        timing = self.timing = {"busy_s": 0.0, "blocked_by_lag_s": 0.0, "wall_s": 0.0, "groups": 0, "versions_seen": 1}
        t_start = time.perf_counter()
        start = self.published_samples
        with write_to_streams(self.data_stream) as writer:
            while True:
                if ts.samples_processed is not None and ts.samples_processed >= samples_target:
                    logger.info("Trainer signalled completion; stopping actor loop")
                    break
                if max_groups is not None and submitted >= max_groups:
                    break
                if ts.propagated_weight_version > last_version:
                    if per_update is not None:
                        can_submit += per_update
                    last_version = ts.propagated_weight_version
                    timing["versions_seen"] += 1
                room = min(concurrent_groups, can_submit - submitted, (max_groups - submitted) if max_groups is not None else concurrent_groups)
                if room <= 0:  # blocked_by_lag (actor.py:556)
                    t0 = time.perf_counter()
                    await asyncio.sleep(poll_s)
                    timing["blocked_by_lag_s"] += time.perf_counter() - t0
                    continue
                t0 = time.perf_counter()
                todo = []
                for k in range(int(room)):
                    i = submitted + k
                    todo.append(({**problems[i % len(problems)], "epoch": i // len(problems)}, i))
                groups = await asyncio.gather(*[self.rollout_group(p, gid) for p, gid in todo])
                submitted += len(todo)
                for g in groups:
                    self.publish(writer, g)
                    if on_group is not None:
                        on_group(g)
                timing["busy_s"] += time.perf_counter() - t0
                timing["groups"] += len(groups)
        timing["wall_s"] = time.perf_counter() - t_start
        return self.published_samples - start

    def run_paced(self, problems: Sequence[dict] | None = None, **kw: Any) -> int:
        return asyncio.run(self.run_paced_async(self.load_problems() if problems is None else problems, **kw))
