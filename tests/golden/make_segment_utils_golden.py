"""Golden vectors for the segment-reduction helpers of the reference's loss (pipelinerl/finetune/rl/utils.py:26-92, 106-208:
`mask_sum`, `mask_mean`, `sum_sum`, `mean_sum`, `per_segment_sums`), imported from /root/reference and run on seeded inputs:
values AND autograd gradients (tests/golden/segment_utils.npz).

    python tests/golden/make_segment_utils_golden.py
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference")

from pipelinerl.finetune.rl import utils as ref  # noqa: E402


def main() -> None:
    rng = np.random.default_rng(77)
    out: dict[str, np.ndarray] = {}
    cases = {"packed": [0, 17, 40, 41, 90, 128], "one_segment": [0, 64], "many": list(range(0, 257, 16))}
    for name, bounds in cases.items():
        L = bounds[-1]
        seg = np.concatenate([np.full(b - a, k) for k, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))])[None].astype(np.int64)
        mask = (rng.random((1, L - 1)) < 0.7)
        if name == "packed":
            mask[0, 40:41] = False  # a segment without a single valid token
        a = rng.normal(size=(1, L - 1)).astype(np.float32)
        b = rng.normal(size=(1, L - 1)).astype(np.float32)
        ups = rng.normal(size=(3, len(bounds) - 1)).astype(np.float32)
        ta, tb = torch.tensor(a, requires_grad=True), torch.tensor(b, requires_grad=True)
        lrn, adv, cnt = ref.per_segment_sums(torch.from_numpy(seg), torch.from_numpy(mask), ta, tb)
        (lrn * torch.from_numpy(ups[0])).sum().backward(retain_graph=True)
        g_a = ta.grad.clone()
        (adv * torch.from_numpy(ups[1])).sum().backward()
        out.update({f"{name}/segment_ids": seg, f"{name}/mask": mask, f"{name}/a": a, f"{name}/b": b, f"{name}/upstream": ups,
                    f"{name}/lrn_sum": lrn.detach().numpy(), f"{name}/adv_sum": adv.detach().numpy(), f"{name}/count": cnt.detach().numpy(),
                    f"{name}/grad_a": g_a.numpy(), f"{name}/grad_b": tb.grad.numpy()})
        # sum_sum / mean_sum over the same segments as (start, end) pairs on the shifted axis (rl/__init__.py:165-185)
        segments = [(a_, b_) for a_, b_ in zip(bounds[:-1], bounds[1:])]
        tv = torch.tensor(a, requires_grad=True)
        m = torch.from_numpy(mask)
        ss = ref.sum_sum(tv, m, segments)
        ss.backward()
        out[f"{name}/sum_sum"], out[f"{name}/sum_sum_grad"] = ss.detach().numpy(), tv.grad.numpy().copy()
        tv.grad = None
        ms = ref.mean_sum(tv, m, segments)
        ms.backward()
        out[f"{name}/mean_sum"], out[f"{name}/mean_sum_grad"] = ms.detach().numpy(), tv.grad.numpy().copy()
        out[f"{name}/sum_sum_unpacked"] = ref.sum_sum(torch.from_numpy(a), m, None).numpy()
        out[f"{name}/mean_sum_unpacked"] = ref.mean_sum(torch.from_numpy(a), m, None).numpy()
        out[f"{name}/mask_sum"] = ref.mask_sum(torch.from_numpy(a), m).numpy()
        out[f"{name}/mask_mean"] = ref.mask_mean(torch.from_numpy(a), m).numpy()
        out[f"{name}/bounds"] = np.asarray(bounds, dtype=np.int64)
    np.savez_compressed(HERE / "segment_utils.npz", **out)
    print("wrote", HERE / "segment_utils.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
