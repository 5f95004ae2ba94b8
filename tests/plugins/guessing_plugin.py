"""A rollout / dataset plugin pair with the reference's plugin signatures (actor.py:141, 803-808), modelled on its
canonical example `pipelinerl/domains/guessing/guessing.py`: guess a number in 1..1024 from "higher / lower" feedback,
up to 13 turns, one training text per turn, reward 2 - turns/10 on success and -2 + turns/10 for a malformed answer.

Test infrastructure: the llm is scripted (`ScriptedLLM`), so the whole plugin surface can be driven without vLLM.
`load_problems` must return what the reference's loader returns - pinned by tests/golden/guessing_problems.json,
which tests/golden/make_guessing_golden.py wrote by calling the reference's own function."""

from __future__ import annotations

import re
import time

from pipelinerl_amd.rollouts import BaseMetrics, RolloutResult, TrainingText

DOMAIN = "guessing"
UPPER, STRIDE, PER_SPLIT, MAX_TURNS = 1024, 191, 512, 13


def load_problems(dataset_names: list[str], **_params) -> list[dict]:
    """Train problems sit on the even multiples of the stride, test problems on the odd ones (mod 1024, 1-based)."""
    out = []
    for name in dataset_names:
        offset = {"train": 0, "test": 1}.get(name)
        if offset is None:
            continue
        out += [{"answer": ((2 * i + offset) * STRIDE) % UPPER + 1, "dataset": name, "domain": DOMAIN} for i in range(PER_SPLIT)]
    return out


class ScriptedLLM:
    """Plays bisection from the feedback lines of the conversation; the calls numbered in `flaky_calls` (or every
    call with `always_fail`) raise a retryable TimeoutError, `malformed_after` makes it forget the answer tags from
    that turn on."""

    def __init__(self, vocab: int = 64, flaky_calls: tuple = (), always_fail: bool = False, malformed_after: int | None = None,
                 eos_token_id: int = 2, split: float = 0.5):
        """`split`: where inside the remaining interval the next guess lands (0.5 = bisection; another value needs more turns,
        so two differently scripted llms give the rollouts of one group different rewards)."""
        self.split = split
        self.vocab, self.flaky_calls, self.always_fail = vocab, set(flaky_calls), always_fail
        self.malformed_after, self.eos = malformed_after, eos_token_id
        self.calls = 0

    def _ids(self, text: str) -> list[int]:
        return [3 + (sum(map(ord, w)) % (self.vocab - 3)) for w in text.split()]

    def generate(self, messages: list[dict]) -> dict:
        self.calls += 1
        if self.always_fail or self.calls in self.flaky_calls:
            raise TimeoutError("scripted time-out")
        lo, hi = 1, UPPER
        turns = 0
        for m in messages[2:]:
            for guess, relation in re.findall(r"(\d+), which is (lower|higher)", m["content"]):
                turns += 1
                lo, hi = (max(lo, int(guess) + 1), hi) if relation == "lower" else (lo, min(hi, int(guess) - 1))
        guess = lo + int((hi - lo) * self.split)
        text = f"my guess {guess}" if (self.malformed_after is not None and turns >= self.malformed_after) else f"I think <answer>{guess}</answer>"
        prompt_ids = self._ids(" ".join(m["content"] for m in messages))
        out_ids = self._ids(text) + [self.eos]
        return {"text": text, "prompt_ids": prompt_ids, "output_ids": out_ids,
                "logprobs": [-0.05 * (1 + (t % 7)) for t in out_ids], "finished": True}


def make_training_text(call: dict, prompt_text: str) -> TrainingText:
    ids = call["prompt_ids"] + call["output_ids"]
    return TrainingText(text=prompt_text + call["text"], n_predicted=len(call["text"]), input_ids=ids,
                        labels=[-100] * len(call["prompt_ids"]) + call["output_ids"], logprobs=call["logprobs"],
                        finished=call["finished"], prompt_tokens=len(call["prompt_ids"]), output_tokens=len(call["output_ids"]))


async def generate_guessing_rollout(cfg, llm, problem: dict, session) -> RolloutResult:
    opening = [{"role": "system", "content": "You are a helpful assistant"},
               {"role": "user", "content": f"Guess a number between 1 and {UPPER}; answer as <answer>number</answer>, I will say higher or lower."}]
    t0 = time.time()
    texts, history = [], []
    reward, success, error = 0.0, False, False
    for turn in range(MAX_TURNS):
        messages = list(opening)
        if history:
            feedback = "\n".join(f"{g}, which is {'lower' if g < problem['answer'] else 'higher'} than the target number." for g in history)
            messages.append({"role": "user", "content": f"Your {turn} previous guesses:\n{feedback}"})
        call = llm.generate(messages)
        texts.append(make_training_text(call, " ".join(m["content"] for m in messages)))
        found = re.search(r"<answer>(\d+)</answer>", call["text"])
        if not found:
            reward, error = -2 + turn / 10, True
            break
        if int(found.group(1)) == problem["answer"]:
            reward, success = 2 - turn / 10, True
            break
        history.append(int(found.group(1)))
    for t in texts:
        t.reward = reward
    return RolloutResult(training_texts=texts, metrics=BaseMetrics(reward=reward, success=success, no_error=not error, no_answer=error),
                         latency=time.time() - t0, dataset_name=problem["dataset"], domain=DOMAIN)


def sync_rollout(cfg, llm, problem: dict, session) -> RolloutResult:
    """A policy that returns its result without being a coroutine (accepted like in domains/dispatcher.py:84-86)."""
    coro = generate_guessing_rollout(cfg, llm, problem, session)  # never suspends: the scripted llm is synchronous
    try:
        coro.send(None)
    except StopIteration as done:
        return done.value
    raise RuntimeError("the scripted rollout suspended")
