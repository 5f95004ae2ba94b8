"""Oracle for the preprocess path: pure-python / numpy restatement of

  prepare_rl_fields      pipelinerl/finetune/rl/__init__.py:573-594
  preprocess_fn          pipelinerl/finetune/data.py:111-160   (token-id branch)
  populate_rl_data       pipelinerl/finetune/rl/__init__.py:453-570
  collate_packed         pipelinerl/finetune/data.py:215-283
  collate                pipelinerl/finetune/data.py:163-212
  create_sentinel_*      pipelinerl/finetune/utils.py:17-78

Written with dict grouping and numpy instead of the reference's pandas / torch code; outputs
use numpy arrays with the reference's dtypes (int64 / float32, seq_boundaries int32).
"""

from __future__ import annotations

import math
from collections import OrderedDict
from typing import Any

import numpy as np

RL_COLUMNS = ("overflow", "group_tokens", "num_labels", "rewards", "advantages", "old_logprobs", "ref_logprobs")


def prepare_fields(entry: dict[str, Any]) -> dict[str, Any]:
    """preprocess_fn(is_rl=True) + prepare_rl_fields for an entry that carries token ids."""
    ids, labels = entry["input_ids"], entry["labels"]
    n = len(labels)
    old, ref = entry["logprobs"], entry["ref_logprobs"]
    assert sum(1 for x in labels if x != -100) == len(old)
    return {
        "input_ids": ids,
        "labels": labels,
        "attention_mask": [1] * n,
        "rewards": [entry["reward"]] * n,
        "advantages": [0.0] * n,
        "old_logprobs": [0] * (n - len(old)) + list(old),
        "ref_logprobs": [0] * (n - len(ref)) + list(ref),
        "overflow": [0] * n,
        "group_tokens": [0] * n,
        "num_labels": [1 if x != -100 else 0 for x in labels],
    }


def _sample_std(xs: list[float]) -> float:
    """pandas groupby std (ddof = 1); NaN for fewer than two values."""
    n = len(xs)
    if n < 2:
        return math.nan
    mean = math.fsum(xs) / n
    return math.sqrt(math.fsum((x - mean) ** 2 for x in xs) / (n - 1))


def sequence_scalars(dataset: list[dict[str, Any]], eos_token_id: int, divide_advantage_by_std: bool):
    """The four per-sequence scalars populate_rl_data broadcasts to per-token lists:
    (advantage, group_tokens, overflow, num_labels), float64 like the reference."""
    by_key: dict[Any, list[float]] = OrderedDict()
    rollout_tokens: dict[Any, int] = OrderedDict()
    for e in dataset:
        by_key.setdefault((e["group_id"], e["step_index"]), []).append(e["rewards"][0])
        rk = (e["group_id"], e["rollout_index"])
        rollout_tokens[rk] = rollout_tokens.get(rk, 0) + len(e["input_ids"])
    group_rollouts: dict[Any, list[int]] = OrderedDict()
    for (gid, _), ntok in rollout_tokens.items():
        group_rollouts.setdefault(gid, []).append(ntok)

    out = []
    for e in dataset:
        rs = by_key[(e["group_id"], e["step_index"])]
        r = e["rewards"][0]
        n = len(rs)
        loo = (math.fsum(rs) - r) / (n - 1) if n > 1 else r
        if divide_advantage_by_std:
            sd = _sample_std(rs)
            adv = (r - loo) / ((0.0 if math.isnan(sd) else sd) + 1e-4)
        else:
            adv = r - loo
        toks = group_rollouts[e["group_id"]]
        gt = sum(toks) / len(toks)
        fr = e.get("finish_reason")
        if isinstance(fr, str) and fr.strip().lower() == "length":
            ovf = 1.0
        elif isinstance(fr, str) and fr.strip().lower() in ("stop", "content_filter"):
            ovf = 0.0
        elif e.get("finished"):
            ovf = 0.0
        else:
            ovf = 0.0 if eos_token_id in e["input_ids"] else 1.0
        nl = sum(1 for x in e["labels"] if x != -100)
        out.append((adv, gt, ovf, nl))
    return out


def populate(dataset: list[dict[str, Any]], eos_token_id: int, divide_advantage_by_std: bool) -> list[dict[str, Any]]:
    """populate_rl_data: per-token lists of the broadcast scalars, in place."""
    for e, (adv, gt, ovf, nl) in zip(dataset, sequence_scalars(dataset, eos_token_id, divide_advantage_by_std)):
        n = len(e["input_ids"])
        e["advantages"] = [adv] * n
        e["group_tokens"] = [gt] * n
        e["overflow"] = [ovf] * n
        e["num_labels"] = [nl] * n
    return dataset


def preprocess_chunk(raw: list[dict[str, Any]], eos_token_id: int, divide_advantage_by_std: bool) -> list[dict[str, Any]]:
    """preprocess_dataset without OOV patching / reference LLM (preprocess.py:145-189)."""
    data = []
    for e in raw:
        e = dict(e)
        if not e.get("ref_logprobs"):
            e["ref_logprobs"] = e["logprobs"]
        e.update(prepare_fields(e))
        meta = e.get("metadata", {})
        e["model_version"] = meta.get("model_version", 0)
        e["rollout_index"] = meta.get("rollout_index", 0)
        e["step_index"] = meta.get("step_index", 0)
        data.append(e)
    return populate(data, eos_token_id, divide_advantage_by_std)


def sentinel_example(n_tokens: int, eos_token_id: int, model_version: int = 0) -> dict[str, Any]:
    rep = lambda v: [v] * n_tokens  # noqa: E731
    return {"input_ids": rep(eos_token_id), "attention_mask": rep(1), "labels": rep(-100),
            "position_ids": list(range(n_tokens)), "rewards": rep(0.0), "advantages": rep(0.0),
            "ref_logprobs": rep(0.0), "old_logprobs": rep(0.0), "group_tokens": rep(1.0), "num_labels": rep(1.0),
            "overflow": rep(0.0), "model_version": model_version}


def sentinel_batch(eos_token_id: int = 2, model_version: int = 0) -> dict[str, Any]:
    n = 8
    i64 = lambda v: np.full((1, n), v, dtype=np.int64)  # noqa: E731
    f32 = lambda v: np.full((1, n), v, dtype=np.float32)  # noqa: E731
    return {"input_ids": i64(eos_token_id), "attention_mask": i64(1), "labels": i64(-100),
            "position_ids": np.arange(n, dtype=np.int64)[None], "segment_ids": i64(0), "rewards": f32(0), "advantages": f32(0),
            "ref_logprobs": f32(0), "old_logprobs": f32(0), "group_tokens": f32(1), "num_labels": f32(1), "overflow": f32(0),
            "seq_boundaries": np.array([0, n], dtype=np.int32), "model_version": model_version, "sentinel": True,
            "is_packed": True, "padding": 0}


def collate_packed(examples: list[dict[str, Any]], eos_token_id: int, seq_parallel: int) -> dict[str, Any]:
    total = sum(len(e["input_ids"]) for e in examples)
    padding = 0
    if total % seq_parallel:
        padding = seq_parallel - total % seq_parallel
        examples = examples + [sentinel_example(padding, eos_token_id, max(e["model_version"] for e in examples))]
    lens = [len(e["input_ids"]) for e in examples]
    bounds = np.zeros(len(examples) + 1, dtype=np.int32)
    bounds[1:] = np.cumsum(lens)
    T = int(bounds[-1])
    out: dict[str, Any] = {
        "input_ids": np.concatenate([np.asarray(e["input_ids"], dtype=np.int64) for e in examples]).reshape(1, T),
        "attention_mask": np.ones((1, T), dtype=np.int64),
        "position_ids": np.concatenate([np.arange(n, dtype=np.int64) for n in lens]).reshape(1, T),
        "segment_ids": np.repeat(np.arange(len(examples), dtype=np.int64), lens).reshape(1, T),
    }
    lab = []
    for i, e in enumerate(examples):
        a = np.asarray(e["labels"], dtype=np.int64).copy()
        if i > 0 and len(a):
            a[0] = -100
        lab.append(a)
    out["labels"] = np.concatenate(lab).reshape(1, T)
    for col in RL_COLUMNS:
        if col in examples[0]:
            out[col] = np.concatenate([np.asarray(e[col], dtype=np.float64) for e in examples]).astype(np.float32).reshape(1, T)
    out.update(model_version=min(e.get("model_version", 0) for e in examples), is_packed=True, seq_boundaries=bounds,
               padding=padding, sentinel=False)
    return out


def collate(examples: list[dict[str, Any]], padding_side: str = "right", pad_to_multiple_of: int = 16) -> dict[str, Any]:
    longest = max(len(e["input_ids"]) for e in examples)
    if longest % pad_to_multiple_of:
        longest += pad_to_multiple_of - longest % pad_to_multiple_of
    out: dict[str, Any] = {}
    spec = {"input_ids": (np.int64, 0), "attention_mask": (np.int64, 0), "labels": (np.int64, -100)}
    spec.update({c: (np.float32, 0.0) for c in RL_COLUMNS})
    for key, (dtype, pad) in spec.items():
        if key not in examples[0]:
            continue
        rows = np.full((len(examples), longest), pad, dtype=np.float64 if dtype == np.float32 else dtype)
        for i, e in enumerate(examples):
            v = e[key]
            if padding_side == "right":
                rows[i, : len(v)] = v
            else:
                rows[i, longest - len(v):] = v
        out[key] = rows.astype(dtype)
    out.update(model_version=min(e.get("model_version", 0) for e in examples), is_packed=False, sentinel=False, padding=0)
    return out
