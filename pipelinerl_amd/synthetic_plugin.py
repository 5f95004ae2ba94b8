"""A rollout / dataset plugin pair that emits the seeded synthetic rollouts of SURVEY.md §8(d) through the reference's
plugin surface (`load_problems(dataset_names, **params) -> list[dict]`, `rollout_policy(cfg, llm, problem, session) ->
RolloutResult`; reference actor.py:141, 803-808), so the pipeline of BASELINE `configs[1]` ("synthetic rollouts bs=512
seq=2048") runs actor -> preprocessor -> learner without an inference server:

    prompt length   P ~ U{prompt_min..prompt_max};  completion C ~ U{seq_length/4 .. seq_length-P} (dense: C = seq_length-P)
    input_ids ~ U{3..vocab-1};  labels = [-100]*P + input_ids[P:];  logprobs = -|N(0,1)| * 0.7
    reward ~ Bernoulli(0.5) per rollout;  finished = C < seq_length-P, a finished rollout ends with EOS

Every draw comes from `numpy.random.default_rng([seed, problem id, attempt])`: the records of a run are a pure function of
the configuration, whatever the scheduling of the stages - which is what lets a test re-derive them for the oracle.
"""

from __future__ import annotations

import time
from typing import Any

import numpy as np

from .rollouts import BaseMetrics, RolloutResult, TrainingText

EOS_TOKEN_ID = 2


def load_problems(dataset_names: list[str], n_problems: int = 64, seed: int = 1235, **_params: Any) -> list[dict]:
    """`n_problems` problems per dataset name; a problem is only an id - the scripted llm derives everything from it."""
    return [{"id": k, "dataset": name, "seed": int(seed)} for name in dataset_names for k in range(int(n_problems))]


class SyntheticLLM:
    """The scripted "inference server": `complete(problem)` returns one rollout; the k-th call for a problem is its k-th attempt."""

    def __init__(self, vocab: int, seq_length: int, prompt_min: int = 64, prompt_max: int = 512, dense: bool = False,
                 eos_token_id: int = EOS_TOKEN_ID, with_ref: bool = False, latency_s: float = 0.0):
        self.vocab, self.seq_length = int(vocab), int(seq_length)
        self.prompt_max = min(int(prompt_max), max(int(prompt_min), self.seq_length - 2))
        self.prompt_min = min(int(prompt_min), self.prompt_max)
        self.dense, self.eos, self.with_ref, self.latency_s = bool(dense), int(eos_token_id), bool(with_ref), float(latency_s)
        self._attempt: dict[tuple, int] = {}
        self.calls = 0

    def complete(self, problem: dict) -> dict:
        key = (problem.get("dataset"), int(problem["id"]), int(problem.get("epoch", 0)))
        attempt = self._attempt.get(key, 0)
        self._attempt[key] = attempt + 1
        self.calls += 1
        return synthetic_rollout(int(problem.get("seed", 0)), int(problem["id"]), attempt, self.vocab, self.seq_length, self.prompt_min,
                                 self.prompt_max, self.dense, self.eos, self.with_ref, epoch=int(problem.get("epoch", 0)))


def synthetic_rollout(seed: int, problem_id: int, attempt: int, vocab: int, seq_length: int, prompt_min: int = 64, prompt_max: int = 512,
                      dense: bool = False, eos_token_id: int = EOS_TOKEN_ID, with_ref: bool = False, epoch: int = 0) -> dict:
    """One rollout as numpy arrays; a pure function of its arguments."""
    rng = np.random.default_rng([int(seed), int(epoch), int(problem_id), int(attempt)])
    prompt_max = min(int(prompt_max), max(int(prompt_min), seq_length - 2))
    prompt_min = min(int(prompt_min), prompt_max)
    P = int(rng.integers(prompt_min, prompt_max + 1))
    cmax = seq_length - P
    C = cmax if dense else min(cmax, min(max(seq_length // 4, 1), cmax) + int(rng.random() * (cmax - min(max(seq_length // 4, 1), cmax) + 1)))
    ids = rng.integers(3, vocab, size=P + C, dtype=np.int64)
    finished = C < cmax
    if finished:
        ids[-1] = eos_token_id
    logprobs = (-np.abs(rng.standard_normal(C)) * 0.7).astype(np.float32)
    ref = (logprobs + rng.standard_normal(C).astype(np.float32) * np.float32(0.05)).astype(np.float32) if with_ref else None
    reward = float(rng.random() < 0.5)
    return {"input_ids": ids, "prompt_len": P, "logprobs": logprobs, "ref_logprobs": ref, "reward": reward, "finished": bool(finished)}


def training_text_of(r: dict) -> TrainingText:
    ids = r["input_ids"].tolist()
    P = int(r["prompt_len"])
    return TrainingText(text="", n_predicted=0, reward=r["reward"], logprobs=r["logprobs"].tolist(),
                        ref_logprobs=r["ref_logprobs"].tolist() if r["ref_logprobs"] is not None else [],
                        input_ids=ids, labels=[-100] * P + ids[P:], finished=r["finished"], prompt_tokens=P, output_tokens=len(ids) - P)


async def generate_rollout(cfg: Any, llm: SyntheticLLM, problem: dict, session: Any) -> RolloutResult:
    """The rollout policy: one single-turn rollout of `problem` on the scripted llm."""
    t0 = time.time()
    if llm.latency_s > 0:
        import asyncio

        await asyncio.sleep(llm.latency_s)
    r = llm.complete(problem)
    text = training_text_of(r)
    return RolloutResult(training_texts=[text], metrics=BaseMetrics(reward=r["reward"], success=r["reward"] > 0, no_error=True, no_answer=False),
                         latency=time.time() - t0, dataset_name=problem.get("dataset"))
