"""Batch ABI of the learner: `PipelineBatchEncoding` and `TrainingMetrics`.

Same field names, dtypes and methods as reference pipelinerl/finetune/types.py:26-180 (the
records on the `training_data` stream are `model_dump()`s of this object), implemented as a
slotted plain class instead of a pydantic model: batches are built from device buffers that
the pack kernel has just written, so there is nothing to validate per field beyond a dtype
coercion, and construction stays O(#fields) instead of O(#tokens).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Iterable

import numpy as np
import torch

_LONG_FIELDS = ("input_ids", "attention_mask", "labels", "position_ids", "segment_ids", "image_grid_thw")
_FLOAT_FIELDS = (
    "rewards",
    "advantages",
    "ref_logprobs",
    "old_logprobs",
    "group_tokens",
    "num_labels",
    "overflow",
    "pixel_values",
)
_INT_FIELDS = ("seq_boundaries",)
_REQUIRED = ("input_ids", "attention_mask", "labels") + _FLOAT_FIELDS[:7] + ("model_version",)
_SCALAR_DEFAULTS = {"sentinel": False, "padding": 0, "is_packed": False}
_OPTIONAL_TENSORS = ("position_ids", "segment_ids", "seq_boundaries", "pixel_values", "image_grid_thw")

# order matters: model_dump() / stream records list fields in the reference's declaration order
_FIELD_ORDER = (
    "input_ids",
    "attention_mask",
    "labels",
    "position_ids",
    "segment_ids",
    "rewards",
    "advantages",
    "ref_logprobs",
    "old_logprobs",
    "group_tokens",
    "num_labels",
    "overflow",
    "model_version",
    "sentinel",
    "padding",
    "is_packed",
    "seq_boundaries",
    "pixel_values",
    "image_grid_thw",
)

# the [*, L] tensors that make_slices() cuts along the token axis
_TOKEN_AXIS_FIELDS = (
    "input_ids",
    "attention_mask",
    "labels",
    "position_ids",
    "segment_ids",
    "rewards",
    "advantages",
    "ref_logprobs",
    "old_logprobs",
    "group_tokens",
    "overflow",
    "num_labels",
)


def _coerce(name: str, value: Any, dtype: torch.dtype) -> torch.Tensor | None:
    if value is None:
        return None
    if isinstance(value, torch.Tensor):
        return value if value.dtype == dtype else value.to(dtype)
    if isinstance(value, (list, tuple, np.ndarray)):
        return torch.as_tensor(np.asarray(value), dtype=dtype) if isinstance(value, np.ndarray) else torch.tensor(value, dtype=dtype)
    raise ValueError(f"Unsupported type for field {name!r}: {type(value)}")


@dataclass
class TrainingMetrics:
    """Counters persisted with checkpoints (reference types.py:26-43)."""

    epoch: int = 0
    passes: int = 0
    completed_steps: int = 0
    samples: int = 0
    tokens: int = 0
    samples_too_old_to_queue: int = 0
    samples_too_old_to_train: int = 0
    last_broadcasted_version: int = 0
    train_loss: float = 1e9
    eval_loss: float = 1e9
    dev_loss: float = 1e9
    grad_norm: float = 0.0
    best_eval_loss: float = 1e9
    best_completed_steps: int = 0
    lr: float = 0.0
    time_waiting_for_data: float = 0.0


class PipelineBatchEncoding:
    """One micro-batch: int64 / float32 tensors of shape [B, L] (packed: [1, T])."""

    __slots__ = _FIELD_ORDER + ("model_extra",)
    model_fields = {name: None for name in _FIELD_ORDER}  # pydantic-compatible introspection

    def __init__(self, **data: Any):
        missing = [k for k in _REQUIRED if k not in data or data[k] is None]
        if missing:
            raise ValueError(f"PipelineBatchEncoding is missing required fields: {missing}")
        for name in _LONG_FIELDS:
            object.__setattr__(self, name, _coerce(name, data.get(name), torch.long))
        for name in _FLOAT_FIELDS:
            object.__setattr__(self, name, _coerce(name, data.get(name), torch.float32))
        for name in _INT_FIELDS:
            object.__setattr__(self, name, _coerce(name, data.get(name), torch.int32))
        self.model_version = int(data["model_version"])
        for name, default in _SCALAR_DEFAULTS.items():
            value = data.get(name, default)
            object.__setattr__(self, name, type(default)(value))
        self.model_extra = {}  # unknown keys are ignored, like the reference's pydantic model

    # -- pydantic-style helpers used by the stream writer / tests ------------------------
    def model_dump(self) -> dict[str, Any]:
        return {name: getattr(self, name) for name in _FIELD_ORDER}

    def tensors(self) -> Iterable[tuple[str, torch.Tensor]]:
        for name in _FIELD_ORDER:
            v = getattr(self, name)
            if isinstance(v, torch.Tensor):
                yield name, v

    def to_device(self, device: str | torch.device) -> "PipelineBatchEncoding":
        for name, t in list(self.tensors()):
            setattr(self, name, t.to(device, non_blocking=True))
        return self

    @classmethod
    def from_dict(cls, data: dict[str, Any], **defaults: Any) -> "PipelineBatchEncoding":
        merged = {**defaults, **data}
        known = {k: v for k, v in merged.items() if k in cls.model_fields}
        inst = cls(**known)
        inst.model_extra.update({k: v for k, v in merged.items() if k not in cls.model_fields})
        return inst

    def make_slices(self, num_slices: int) -> list["PipelineBatchEncoding"]:
        """Cut a packed batch into `num_slices` equal token ranges (sequence parallelism)."""
        if self.position_ids is None or self.input_ids.shape[0] > 1:
            raise ValueError("Cannot a batch that is not properly packed")
        total = self.input_ids.shape[1]
        if total < num_slices:
            raise ValueError(f"Cannot slice batch of size {total} into {num_slices} slices")
        if total % num_slices != 0:
            raise ValueError(f"Sequence length {total} is not divisible by number of slices {num_slices}")
        bounds = [i * total // num_slices for i in range(num_slices + 1)]
        shared = {
            k: getattr(self, k)
            for k in ("model_version", "sentinel", "is_packed", "padding", "seq_boundaries", "pixel_values", "image_grid_thw")
        }
        out = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            piece = dict(shared)
            for name in _TOKEN_AXIS_FIELDS:
                t = getattr(self, name)
                piece[name] = None if t is None else t[:, lo:hi]
            out.append(PipelineBatchEncoding(**piece))
        return out

    def __repr__(self) -> str:
        shape = tuple(self.input_ids.shape)
        return (
            f"PipelineBatchEncoding(shape={shape}, device={self.input_ids.device}, packed={self.is_packed}, "
            f"sentinel={self.sentinel}, model_version={self.model_version})"
        )
