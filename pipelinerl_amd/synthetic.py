"""Seeded synthetic rollouts: the same generator feeds the HIP path, the
oracle and the golden-vector script, so parity tests and the benchmark see identical inputs.

    groups of `attempts` rollouts, group_id "g<k>", step_index 0, model_version 0
    prompt length   P ~ U{prompt_min..prompt_max}
    completion      C ~ U{seq_length/4 .. seq_length-P}   (dense=True: C = seq_length-P)
    input_ids ~ U{3..vocab-1};  labels = [-100]*P + input_ids[P:]
    logprobs  = -|N(0,1)| * 0.7;  ref_logprobs = logprobs + N(0, 0.05) when with_ref
    reward ~ Bernoulli(0.5) per rollout (some groups end up with zero variance)
    finished = C < seq_length-P; an EOS token closes finished rollouts;
    finish_reason present for every other rollout ("stop"/"length"), absent otherwise
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .ragged import RaggedRollouts

EOS_TOKEN_ID = 2


def make_ragged(
    n_groups: int,
    attempts: int = 8,
    seq_length: int = 8192,
    vocab: int = 152064,
    seed: int = 1234,
    prompt_min: int = 64,
    prompt_max: int = 512,
    dense: bool = False,
    with_ref: bool = False,
    eos_token_id: int = EOS_TOKEN_ID,
) -> tuple[RaggedRollouts, list[Any]]:
    """Returns (rollouts, finish_reasons) where finish_reasons[i] is a str or None."""
    rng = np.random.default_rng(seed)
    n = n_groups * attempts
    prompt_max = min(prompt_max, max(prompt_min, seq_length - 2))
    prompt_min = min(prompt_min, prompt_max)
    P = rng.integers(prompt_min, prompt_max + 1, size=n, dtype=np.int64)
    cmax = seq_length - P
    if dense:
        C = cmax.copy()
    else:
        cmin = np.minimum(np.maximum(seq_length // 4, 1), cmax)
        C = cmin + (rng.random(n) * (cmax - cmin + 1)).astype(np.int64)
        C = np.minimum(C, cmax)
    L = P + C
    seq_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(L, out=seq_off[1:])
    lp_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(C, out=lp_off[1:])
    total, total_c = int(seq_off[-1]), int(lp_off[-1])

    tokens = rng.integers(3, vocab, size=total, dtype=np.int32)
    # position of each token inside its sequence -> prompt mask
    seq_of_tok = np.repeat(np.arange(n, dtype=np.int64), L)
    pos = np.arange(total, dtype=np.int64) - seq_off[:-1][seq_of_tok]
    is_prompt = pos < P[seq_of_tok]
    finished = (C < cmax).astype(np.uint8)
    # finished rollouts end with EOS
    last = seq_off[1:] - 1
    tokens[last[finished == 1]] = eos_token_id
    labels = np.where(is_prompt, np.int32(-100), tokens).astype(np.int32)

    logprobs = (-np.abs(rng.standard_normal(total_c)) * 0.7).astype(np.float32)
    ref = None
    if with_ref:
        ref = (logprobs + rng.standard_normal(total_c).astype(np.float32) * np.float32(0.05)).astype(np.float32)
    reward = (rng.random(n) < 0.5).astype(np.float64)

    group_index = np.repeat(np.arange(n_groups, dtype=np.int32), attempts)
    rollout_index = np.tile(np.arange(attempts, dtype=np.int32), n_groups)
    step_index = np.zeros(n, dtype=np.int32)
    model_version = np.zeros(n, dtype=np.int64)
    has_reason = (np.arange(n) % 2) == 0
    finish_reasons: list[Any] = [
        (("stop" if finished[i] else "length") if has_reason[i] else None) for i in range(n)
    ]
    finish_code = np.array(
        [0 if r is None else (2 if r == "stop" else 1) for r in finish_reasons], dtype=np.uint8
    )
    rag = RaggedRollouts.from_numpy(
        tokens, labels, logprobs, ref, seq_off, lp_off, reward, group_index, step_index, rollout_index,
        model_version, finished, finish_code, group_ids=[f"g{g}" for g in range(n_groups)],
    )
    return rag, finish_reasons


def ragged_to_entries(rag: RaggedRollouts, finish_reasons: list[Any] | None = None) -> list[dict[str, Any]]:
    """The same rollouts as `actor`-stream dicts (what the reference's preprocessor reads); the
    synthetic `finish_reasons` (mixed case, absent keys) replace the canonical strings."""
    if finish_reasons is None:
        finish_reasons = [None] * rag.n_seqs
    return rag.to_entries(finish_reasons)


def make_entries(n_groups: int, **kw: Any) -> list[dict[str, Any]]:
    rag, reasons = make_ragged(n_groups, **kw)
    return ragged_to_entries(rag, reasons)
