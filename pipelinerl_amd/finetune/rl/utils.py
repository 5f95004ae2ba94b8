"""Helpers of the reference's loss module (pipelinerl/finetune/rl/utils.py): the per-step aggregation of the stats dicts (:9-23) and
the masked segment reductions (:26-92, 106-208).

`rl_step` itself never calls the reductions here - they live inside the HIP loss kernels (prl_loss.hip).  They are provided with
the reference's names and signatures for code that builds its own loss terms out of them: `per_segment_sums`, `sum_sum` and
`mean_sum` run on the fixed-order segmented reduction kernel (`prl_segment_sums`) and are differentiable like the reference's;
`mask_sum` / `mask_mean` are the two one-line tensor expressions they are in the reference.  Device tensors only.
"""

from __future__ import annotations

import math
from typing import Any, Iterable, Mapping

import torch


def _rule(key: str):
    # the aggregation op is chosen by substring, checked in this order (reference :12-21)
    if "min" in key:
        return "min"
    if "max" in key:
        return "max"
    if "loss" in key or "sum" in key:
        return "sum"
    return "mean"


def aggregate_rl_stats(rl_stats: Mapping[str, Iterable[float]], num_samples: int) -> dict[str, float]:
    """{key: [per-micro-batch values]} -> {"rl/key": aggregate}.  min/max keys take the extreme,
    loss/sum keys the sum, everything else sum / num_samples; float32 like torch.Tensor(v)."""
    import numpy as np

    out: dict[str, float] = {}
    for key, values in rl_stats.items():
        v = np.asarray(list(values), dtype=np.float32)
        rule = _rule(key)
        if rule == "min":
            r = v.min()
        elif rule == "max":
            r = v.max()
        elif rule == "sum":
            r = v.sum(dtype=np.float32)
        else:
            r = v.sum(dtype=np.float32) / np.float32(num_samples)
        out["rl/" + key] = float(r)
    return out


def effective_sample_size(avg: Mapping[str, float]) -> float:
    """rl/ess as logged by the trainer (reference finetune_loop.py:912-916)."""
    denom = avg["rl/ratio_new_old_squared_sum"] * avg["rl/num_output_tokens_sum"]
    return avg["rl/ratio_new_old_sum"] ** 2 / denom if denom else math.nan


# ---------------------------------------------------------------------------------------------
# masked / segmented reductions (reference :26-92, 106-208)
# ---------------------------------------------------------------------------------------------


def mask_sum(values: torch.Tensor, mask: torch.Tensor, axis: int | None = None) -> torch.Tensor:
    """Sum of the masked values; non-finite products count as 0 (:26-31)."""
    prod = (values * mask).nan_to_num(0)
    return prod.sum() if axis is None else prod.sum(axis=axis)


def mask_mean(values: torch.Tensor, mask: torch.Tensor, axis: int | None = None) -> torch.Tensor:
    """Masked mean with an empty selection counting as one element (:34-42)."""
    if axis is None:
        return mask_sum(values, mask) / mask.sum().clamp(min=1).to(values.dtype)
    return mask_sum(values, mask, axis) / mask.sum(axis=axis).clamp(min=1).to(values.dtype)


class _SegmentSums(torch.autograd.Function):
    """(a, b) fp32 [1, L - 1] on the SHIFTED axis, mask [1, L - 1], token-aligned segment ids int64 [1, L] -> per-segment masked sums of a
    and b and token counts, fp32 [n_segments] x 3 (`prl_segment_sums`: fixed-order fp64 reduction).  Backward: every token takes its
    segment's upstream value, times the mask (times `grad_scale`: the sequence-parallel group size, like a differentiable all-reduce)."""

    @staticmethod
    def forward(ctx, a, b, mask, seg_full, n_segments: int, group, grad_scale: float):  # type: ignore[override]
        from . import _group_all_reduce, segment_sums

        L = seg_full.shape[-1]
        dev = a.device
        valid = mask.reshape(1, -1) != 0
        labels = torch.full((1, L), -100, dtype=torch.int64, device=dev)
        labels[:, 1:] = torch.where(valid, 0, -100)
        pad = lambda t: torch.cat([torch.zeros((1, 1), dtype=torch.float32, device=dev), t.detach().reshape(1, -1).to(torch.float32)], dim=1)  # noqa: E731
        sa, sb, cnt = segment_sums(seg_full, labels, pad(a), pad(b), n_segments) if n_segments else (torch.zeros(0, dtype=torch.float64, device=dev),) * 3
        if group is not None:
            import torch.distributed as dist

            sa, sb, cnt = _group_all_reduce(torch.stack([sa, sb, cnt]), group, dist.ReduceOp.SUM)
        ctx.save_for_backward(seg_full, valid)
        ctx.grad_scale, ctx.shapes, ctx.dtypes = float(grad_scale), (a.shape, b.shape), (a.dtype, b.dtype)
        out_dt = a.dtype
        cnt = cnt.to(out_dt)
        ctx.mark_non_differentiable(cnt)
        return sa.to(out_dt), sb.to(out_dt), cnt

    @staticmethod
    def backward(ctx, g_a, g_b, _g_cnt):  # type: ignore[override]
        seg_full, valid = ctx.saved_tensors
        idx = seg_full[0, 1:].clamp(min=0)
        scale = valid[0].to(torch.float32) * ctx.grad_scale
        ga = (g_a.to(torch.float32)[idx] * scale).reshape(ctx.shapes[0]).to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
        gb = (g_b.to(torch.float32)[idx] * scale).reshape(ctx.shapes[1]).to(ctx.dtypes[1]) if ctx.needs_input_grad[1] else None
        return ga, gb, None, None, None, None, None


def per_segment_sums(segment_ids: torch.Tensor, masks_shifted: torch.Tensor, log_ratio_new_old: torch.Tensor, advantages: torch.Tensor,
                     seq_parallel_group: Any = None) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Differentiable per-segment reductions with the optional sequence-parallel all-reduce (:106-208): `segment_ids` int64 [1, L]
    (token-aligned, non-decreasing), the other three on the shifted axis [1, L - 1].  Returns (sum of log_ratio_new_old, sum of
    advantages, token count) per segment, over the masked tokens; with a group, summed over its ranks (ONE all-reduce of the three
    columns; the number of segments is agreed with one MAX all-reduce first, as in the reference)."""
    import torch.distributed as dist

    from ... import _lib
    from . import _group_all_reduce

    if segment_ids is None:
        raise ValueError("segment_ids must be provided for per-segment reductions")
    if segment_ids.dim() != 2 or segment_ids.shape[0] != 1:
        raise ValueError(f"Expected segment_ids shaped [1, L], got {tuple(segment_ids.shape)}")
    _lib.require_device(segment_ids, log_ratio_new_old, advantages)
    sp = seq_parallel_group is not None and dist.is_available() and dist.is_initialized()
    seg = segment_ids[:, 1:]
    top = seg.max().to(torch.int64).reshape(1) if seg.numel() else torch.full((1,), -1, dtype=torch.int64, device=segment_ids.device)
    if sp:
        top = _group_all_reduce(top, seq_parallel_group, dist.ReduceOp.MAX)
    n_segments = int(top.item()) + 1
    n = seg.shape[-1]
    if masks_shifted.shape[-1] < n or log_ratio_new_old.shape[-1] < n or advantages.shape[-1] < n:
        raise ValueError("Mask shape mismatch after alignment with segment_ids")
    scale = float(dist.get_world_size(seq_parallel_group)) if sp else 1.0
    return _SegmentSums.apply(log_ratio_new_old[:, :n], advantages[:, :n], masks_shifted[:, :n], segment_ids.contiguous().to(torch.int64), n_segments,
                              seq_parallel_group if sp else None, scale)


def _segment_ids_of(segments: list, length: int, device) -> tuple[torch.Tensor, int]:
    """(start, end) pairs on the shifted axis -> token-aligned int64 ids [1, length + 1]; uncovered positions get an extra, ignored id."""
    import numpy as np

    ids = np.full(length + 1, len(segments), dtype=np.int64)
    ids[0] = 0
    for k, (a, b) in enumerate(segments):
        ids[1 + a: 1 + min(b, length)] = k
    if np.any(np.diff(ids[1:]) < 0):
        raise ValueError("segments must be given in ascending order")
    return torch.from_numpy(ids).to(device).unsqueeze(0), len(segments) + 1


def _per_segment(values: torch.Tensor, masks: torch.Tensor, segments: list):
    assert values.shape[0] == 1, "seq packed samples must have dimension 0 of 1"
    seg_full, n = _segment_ids_of(segments, values.shape[-1], values.device)
    sums, _, counts = _SegmentSums.apply(values.nan_to_num(0), torch.zeros_like(values), masks, seg_full, n, None, 1.0)
    return sums[:-1], counts[:-1]


def sum_sum(values: torch.Tensor, masks: torch.Tensor, segments: list | None):
    """Packed (segments given): the masked sum inside every segment, summed over the segments; otherwise the masked sum (:71-92).
    A sentinel batch (`values.shape[-1] == 1`) takes the unpacked form like in the reference."""
    if segments and values.shape[-1] != 1:
        return _per_segment(values, masks, segments)[0].sum()
    return mask_sum(values, masks)


def mean_sum(values: torch.Tensor, masks: torch.Tensor, segments: list | None):
    """Packed: the masked MEAN inside every segment (an empty segment counts one element), summed over the segments; otherwise the
    row-wise masked mean, summed (:45-68)."""
    if segments and values.shape[-1] != 1:
        sums, counts = _per_segment(values, masks, segments)
        return (sums / counts.clamp(min=1)).sum()
    return mask_mean(values, masks, -1).sum()
