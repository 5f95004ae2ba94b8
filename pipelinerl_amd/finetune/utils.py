"""Sentinel (zero-contribution) batches and filler examples.

Every data-parallel rank must run the same number of forward/backward passes per optimizer
step; ranks that already hold their sample quota receive an all-masked 8-token packed batch
(reference pipelinerl/finetune/utils.py:17-78, used at preprocess.py:599-607 and, for
sequence-parallel padding, data.py:222-230).
"""

from __future__ import annotations

from typing import Any

import torch

from .types import PipelineBatchEncoding

SENTINEL_LENGTH = 8


def _eos(tokenizer: Any, default: int = 2) -> int:
    return getattr(tokenizer, "eos_token_id", default) if tokenizer else default


def create_sentinel_batch(device: Any, tokenizer: Any = None, model_version: int = 0) -> PipelineBatchEncoding:
    """An 8-token packed batch with every label masked: finite zero loss, still a full
    forward/backward.  group_tokens = num_labels = 1 keep the divisions finite."""
    n = SENTINEL_LENGTH
    i64 = lambda v: torch.full((1, n), v, dtype=torch.long)  # noqa: E731
    f32 = lambda v: torch.full((1, n), v, dtype=torch.float32)  # noqa: E731
    batch = PipelineBatchEncoding(
        input_ids=i64(_eos(tokenizer)),
        attention_mask=i64(1),
        labels=i64(-100),
        position_ids=torch.arange(n, dtype=torch.long).unsqueeze(0),
        segment_ids=i64(0),
        rewards=f32(0.0),
        advantages=f32(0.0),
        ref_logprobs=f32(0.0),
        old_logprobs=f32(0.0),
        group_tokens=f32(1.0),
        num_labels=f32(1.0),
        overflow=f32(0.0),
        seq_boundaries=torch.tensor([0, n], dtype=torch.int32),
        model_version=model_version,
        sentinel=True,
        is_packed=True,
    )
    return batch.to_device(device) if device is not None else batch


def create_sentinel_example(n_tokens: int, tokenizer: Any = None, model_version: int = 0) -> dict:
    """Filler example of `n_tokens` masked EOS tokens (list form, for list-of-dicts callers)."""
    eos = tokenizer.eos_token_id
    ints = lambda v: [v] * n_tokens  # noqa: E731
    return {
        "input_ids": ints(eos),
        "attention_mask": ints(1),
        "labels": ints(-100),
        "position_ids": list(range(n_tokens)),
        "rewards": ints(0.0),
        "advantages": ints(0.0),
        "ref_logprobs": ints(0.0),
        "old_logprobs": ints(0.0),
        "group_tokens": ints(1.0),
        "num_labels": ints(1.0),
        "overflow": ints(0.0),
        "model_version": model_version,
    }
