"""Metric aggregation helpers (reference pipelinerl/finetune/rl/utils.py:9-23).

The masked segment reductions of the reference (`mask_sum`, `sum_sum`, `per_segment_sums`) live
inside the HIP loss kernel; what remains on the host is the per-step aggregation of the stats
dicts collected over micro-batches and ranks.
"""

from __future__ import annotations

import math
from typing import Iterable, Mapping


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
