"""Shared helpers for the test-suite: golden fixture loading and comparisons."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

RL_STEP_CASES = sorted(p.stem[len("rl_step_"):] for p in GOLDEN.glob("rl_step_*.npz"))
PREPROCESS_CASES = sorted(p.stem[len("preprocess_"):] for p in GOLDEN.glob("preprocess_*.npz"))

BATCH_TENSOR_KEYS = (
    "input_ids", "attention_mask", "labels", "position_ids", "segment_ids", "rewards", "advantages", "ref_logprobs",
    "old_logprobs", "group_tokens", "num_labels", "overflow", "seq_boundaries",
)
INT_KEYS = ("input_ids", "attention_mask", "labels", "position_ids", "segment_ids", "seq_boundaries")


def load_rl_case(name: str) -> dict:
    """`name`: an rl_step case name, or the stem of any fixture with the same layout (gspo_sp2_*)."""
    path = GOLDEN / f"rl_step_{name}.npz"
    if not path.exists():
        path = GOLDEN / f"{name}.npz"
    z = np.load(path, allow_pickle=False)
    batch = {k[len("batch/"):]: z[k] for k in z.files if k.startswith("batch/")}
    for k in ("model_version", "sentinel", "padding", "is_packed"):
        if k in batch:
            batch[k] = batch[k].item()
    stats = dict(zip([str(k) for k in z["stats_keys"]], [float(v) for v in z["stats_values"]]))
    extra = {k: z[k] for k in ("value", "grad_value") if k in z.files}  # value-head cases (c18-c23)
    return {
        **extra,
        "batch": batch,
        "logits": z["logits"],
        "loss": float(z["loss"]),
        "grad_logits": z["grad_logits"],
        "stats": stats,
        "config": json.loads(str(z["config_json"])),
        "steps": tuple(int(x) for x in z["steps"]),
    }


def load_preprocess_case(name: str) -> dict:
    z = np.load(GOLDEN / f"preprocess_{name}.npz", allow_pickle=False)
    meta = json.loads((GOLDEN / f"preprocess_{name}.json").read_text())
    out = {"raw": meta["raw"], "divide_advantage_by_std": meta["divide_advantage_by_std"], "eos_token_id": meta["eos_token_id"],
           "scalars": {k: z[k] for k in ("advantage", "group_tokens", "overflow", "num_labels")}, "packed": {}, "padded": {}}
    for k in z.files:
        parts = k.split("/")
        if parts[0] in ("packed", "padded"):
            out[parts[0]].setdefault(parts[1], {})[parts[2]] = z[k]
    return out


def assert_batch_equal(got: dict, want: dict, float_tol: float = 0.0):
    """Integer fields bit-exact; float fields exact unless a tolerance is given."""
    for k in BATCH_TENSOR_KEYS:
        if k not in want:
            continue
        w = np.asarray(want[k])
        g = np.asarray(got[k])
        assert g.shape == w.shape, f"{k}: shape {g.shape} != {w.shape}"
        assert g.dtype == w.dtype, f"{k}: dtype {g.dtype} != {w.dtype}"
        if k in INT_KEYS or float_tol == 0.0:
            assert np.array_equal(g, w), f"{k} differs"
        else:
            np.testing.assert_allclose(g, w, rtol=float_tol, atol=0, err_msg=k)
    for k in ("model_version", "is_packed", "padding", "sentinel"):
        if k in want:
            assert int(got[k]) == int(np.asarray(want[k]).item() if hasattr(want[k], "item") else want[k]), k


def rel_err(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max() / denom)
