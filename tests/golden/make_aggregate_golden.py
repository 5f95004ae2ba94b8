"""Golden for the per-step metric aggregation: the reference's `aggregate_rl_stats`
(rl/utils.py:9-23) and `linear_decay_coef` (rl/__init__.py:119-133), imported and run on the
per-micro-batch statistics stored in the rl_step_*.npz fixtures (plus synthetic NaN-free extremes).

    python tests/golden/make_aggregate_golden.py
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))


def main() -> None:
    import make_golden as mg

    ref_rl, _, _ = mg.import_reference()
    from pipelinerl.finetune.rl.utils import aggregate_rl_stats

    cases = []
    names = sorted(p.stem for p in HERE.glob("rl_step_*.npz") if "sentinel" not in p.stem)
    for group, num_samples in ((names[:4], 16), (names[4:9], 24), (names, 64), (names[:1], 3)):
        stats: dict[str, list] = {}
        for n in group:
            z = np.load(HERE / f"{n}.npz")
            for k, v in zip(z["stats_keys"], z["stats_values"]):
                stats.setdefault(str(k), []).append(float(v))
        out = aggregate_rl_stats(stats, num_samples)
        cases.append({"stats": stats, "num_samples": num_samples, "out": out})
    decay = [{"args": [c, m, a, b], "out": ref_rl.linear_decay_coef(c, m, a, b)}
             for c, m, a, b in ((0, 10, 0.1, 0.0), (3, 10, 0.3, 0.1), (10, 10, 0.5, 0.0), (7, 9, 0.0, 0.02), (1, 3, 0.05, 0.05))]
    (HERE / "aggregate.json").write_text(json.dumps({"cases": cases, "linear_decay": decay}, indent=1))
    print(len(cases), "cases;", list(cases[0]["out"].items())[:4])


if __name__ == "__main__":
    main()
