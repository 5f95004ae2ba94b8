"""Golden description of the rollout-plugin data models: the reference's `pipelinerl/rollouts.py`
(`BaseMetrics`, `TrainingText`, `RolloutResult`) imported, their pydantic field tables (name, required,
default) and the `model_dump()` of example instances recorded; plus the field table of `RLConfig`
(rl/__init__.py:43-105) -> rlconfig_fields.json.

    python tests/golden/make_rollouts_golden.py
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent

EXAMPLE_TEXT = dict(text="what is 2+2? 4", n_predicted=1, reward=1.0, logprobs=[-0.25], input_ids=[5, 6, 7, 8], labels=[-100, -100, -100, 8],
                    finished=True, prompt_tokens=3, output_tokens=1, metadata={"k": 1})
EXAMPLE_METRICS = dict(reward=1.0, success=True, no_error=True, no_answer=False)


def field_table(model) -> dict:
    out = {}
    for name, f in model.model_fields.items():
        default = None if f.is_required() else (f.default_factory() if f.default_factory is not None else f.default)
        out[name] = {"required": bool(f.is_required()), "default": default}
    return out


def describe(mod) -> dict:
    text = mod.TrainingText(**EXAMPLE_TEXT)
    minimal = mod.TrainingText(text="ab", n_predicted=1)
    result = mod.RolloutResult(training_texts=[text], metrics=mod.BaseMetrics(**EXAMPLE_METRICS), latency=0.5)
    return {
        "fields": {n: field_table(getattr(mod, n)) for n in ("BaseMetrics", "TrainingText", "RolloutResult")},
        "dumps": {"text": text.model_dump(), "minimal": minimal.model_dump(), "result": result.model_dump()},
        "properties": {"prompt_text": text.prompt_text, "output_text": text.output_text},
    }


def main() -> None:
    sys.path.insert(0, "/root/reference")
    from pipelinerl import rollouts

    out = describe(rollouts)
    (HERE / "rollouts_models.json").write_text(json.dumps(out, indent=1))
    # RLConfig (rl/__init__.py:43-105): every yaml key of conf/finetune/*.yaml `rl:` and its default
    sys.path.insert(0, str(HERE))
    import make_golden as mg

    ref_rl, _, _ = mg.import_reference()
    (HERE / "rlconfig_fields.json").write_text(json.dumps(field_table(ref_rl.RLConfig), indent=1))
    print(json.dumps(out["fields"]["RolloutResult"]))


if __name__ == "__main__":
    main()
