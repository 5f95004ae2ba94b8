"""Golden fixture of the dataset plugin: what the REFERENCE's `domains/guessing/guessing.py::load_problems`
returns for ["train"], ["test"], ["train", "test"] and an unknown name.  The module's heavy imports
(`pipelinerl.async_llm`, `pipelinerl.llm`, omegaconf) are stubbed; only `load_problems` runs.

    python tests/golden/make_guessing_golden.py      (needs /root/reference; writes guessing_problems.json)
"""

import json
import sys
import types
from pathlib import Path

REFERENCE = "/root/reference"


def import_reference_guessing():
    sys.path.insert(0, REFERENCE)
    for name, attrs in {"omegaconf": ("DictConfig", "ListConfig", "OmegaConf"), "pipelinerl.async_llm": ("llm_async_generate", "make_training_text"),
                        "pipelinerl.llm": ("Prompt", "TrainableLLM")}.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, type(a, (), {}))
            sys.modules[name] = m
    # load the FILE, not the package: `pipelinerl.domains/__init__.py` pulls in the dispatcher and its dependencies
    import importlib.util

    spec = importlib.util.spec_from_file_location("reference_guessing", f"{REFERENCE}/pipelinerl/domains/guessing/guessing.py")
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    return g


if __name__ == "__main__":
    g = import_reference_guessing()
    full = {"|".join(names): g.load_problems(list(names)) for names in (("train",), ("test",), ("train", "test"), ("other",))}
    # compact form: every record is {"answer": a, "dataset": <split>, "domain": "guessing"} - checked here, stored as answers
    for split in ("train", "test"):
        assert all(p == {"answer": p["answer"], "dataset": split, "domain": g.DOMAIN} for p in full[split])
    assert full["train|test"] == full["train"] + full["test"] and full["other"] == []
    out = {"record_keys": ["answer", "dataset", "domain"], "domain": g.DOMAIN,
           "answers": {split: [p["answer"] for p in full[split]] for split in ("train", "test")},
           "concatenates_in_argument_order": True, "unknown_name_is_ignored": True}
    path = Path(__file__).resolve().parent / "guessing_problems.json"
    path.write_text(json.dumps(out, sort_keys=True))
    print(path, {k: len(v) for k, v in out["answers"].items()})
