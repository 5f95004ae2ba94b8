"""Golden sample accounting: the reference's `get_batch_token_count`, `get_batch_sequence_count`
(finetune_loop.py:295-312) and `calculate_train_steps` (:1100-1107), cut from the source and
`exec`ed (the module itself cannot be imported), applied to batches collated by the reference's own
`collate_packed` / `collate` and its sentinel batch.

    python tests/golden/make_counts_golden.py
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))


def main() -> None:
    import make_golden as mg
    from make_weight_update_golden import REF, cut

    ref_rl, ref_data, ref_utils = mg.import_reference()
    from pipelinerl.finetune.types import PipelineBatchEncoding

    ns = {"PipelineBatchEncoding": PipelineBatchEncoding}
    exec(compile(cut(REF / "finetune_loop.py", "def get_batch_token_count", "def validate_packing_config"), "ref_counts", "exec"), ns)
    lines = (REF / "finetune_loop.py").read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("def calculate_train_steps"))  # last function of the file
    exec(compile("\n".join(lines[start:]), "ref_steps", "exec"), ns)
    out = {"batches": {}, "train_steps": []}
    for name, case in mg.make_raw_cases().items():
        cfg = ref_rl.RLConfig(divide_advantage_by_std=case["divide_advantage_by_std"])
        data = mg.ref_preprocess(ref_rl, ref_data, case["raw"], cfg)
        n = len(data)
        for pname, (idxs, sp) in {"all_sp1": (list(range(n)), 1), "first5_sp4": (list(range(min(5, n))), 4), "tail3_sp1": (list(range(n - 3, n)), 1)}.items():
            b = ref_data.collate_packed([data[i] for i in idxs], mg.Tok(mg.EOS), seq_parallel=sp)
            out["batches"][f"{name}/packed/{pname}"] = {"tokens": int(ns["get_batch_token_count"](b)), "sequences": int(ns["get_batch_sequence_count"](b)),
                                                        "padding": int(b.padding)}
        for side in ("right", "left"):
            exs = [{k: v for k, v in data[i].items() if k != "finish_reason"} for i in range(min(4, n))]
            b = ref_data.collate(exs, mg.Tok(mg.EOS, side))
            out["batches"][f"{name}/padded/{side}"] = {"tokens": int(ns["get_batch_token_count"](b)), "sequences": int(ns["get_batch_sequence_count"](b))}
    s = ref_utils.create_sentinel_batch(device="cpu", tokenizer=mg.Tok(7), model_version=5)
    out["batches"]["sentinel"] = {"tokens": int(ns["get_batch_token_count"](s)), "sequences": int(ns["get_batch_sequence_count"](s))}
    for cfg_interrupt, cfg_max, arg in ((-1, 100, -1), (40, 100, -1), (100, 100, -1), (-1, 100, 30), (40, 100, 100), (0, 100, -1), (-1, 100, 0)):
        args = types.SimpleNamespace(interrupt_train_steps=cfg_interrupt, max_train_steps=cfg_max)
        out["train_steps"].append({"cfg_interrupt": cfg_interrupt, "max": cfg_max, "arg": arg, "result": ns["calculate_train_steps"](args, arg)})
    (HERE / "batch_counts.json").write_text(json.dumps(out, indent=1))
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main()
