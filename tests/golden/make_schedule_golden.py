"""Golden traces of the preprocessor's micro-batch schedule, produced by EXECUTING the reference's
own scheduling loop.

`pipelinerl/preprocess.py` cannot be imported here (litellm / omegaconf / redis are absent), and the
rule lives in the middle of `run_preprocessing_loop` (lines 594-662), not in a function.  This
script therefore reads the reference source at generation time, cuts out exactly that `while` block,
dedents it and `exec`s it in a namespace whose collaborators (collate, stream writer, sentinel
factory, cfg) are recording stubs.  Nothing from the reference is stored in this repository — only
the resulting traces (tests/golden/schedule.json).

    python tests/golden/make_schedule_golden.py
"""

from __future__ import annotations

import json
import textwrap
import types
from collections import deque
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = Path("/root/reference/pipelinerl/preprocess.py")


def reference_loop_source() -> str:
    lines = SRC.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.strip().startswith("while (len(processed_entries_queue) > 0 and not batch_done)"))
    indent = len(lines[start]) - len(lines[start].lstrip())
    end = start + 1
    while end < len(lines) and (not lines[end].strip() or len(lines[end]) - len(lines[end].lstrip()) > indent):
        end += 1
    return textwrap.dedent("\n".join(lines[start:end]))


class Recorder:
    def __init__(self):
        self.trace = []

    def write(self, trainer_id, kind, ids):
        self.trace.append([int(trainer_id), kind, ids])


def run_reference(params: dict, pushes: list[list[int]]) -> list[dict]:
    """Feed the reference loop `pushes` (each a list of sample lengths arriving together) and
    record what it emits after every arrival."""
    src = reference_loop_source()
    code = compile(src, "reference_schedule_loop", "exec")
    rec = Recorder()
    num_trainers = params["num_trainers"]
    sp = params["seq_parallel"]
    cfg = types.SimpleNamespace(
        preprocess=types.SimpleNamespace(dataset_buffer_size=0),
        finetune=types.SimpleNamespace(seq_packing=params["seq_packing"], seq_parallel=sp, seq_length=params["seq_length"],
                                       train_batch_size=params["train_batch_size"]),
    )
    num_lead = num_trainers // sp
    passes_per_lead = params["gradient_accumulation_passes"] // num_lead
    samples_per_lead_per_step = params["train_batch_size"] * passes_per_lead
    published = params.get("published_samples", 0)
    ns = dict(
        cfg=cfg, logger=types.SimpleNamespace(debug=lambda *a, **k: None, info=lambda *a, **k: None),
        processed_entries_queue=deque(), batch_done=False, trainer_id=0, num_trainers=num_trainers,
        samples_per_trainer={i: published // num_trainers for i in range(0, num_trainers, sp)},
        target_samples_per_lead=published // num_trainers + samples_per_lead_per_step,
        samples_per_lead_per_step=samples_per_lead_per_step, train_batch_size=samples_per_lead_per_step * num_lead,
        batch_boundary=published + samples_per_lead_per_step * num_lead, published_samples=published,
        current_batch=[], current_length=0, time_to_write=False, tokenizer=None, max_model_version=0, data_writer=None,
        create_sentinel_batch=lambda **kw: ("sentinel", []),
        collate_packed=lambda batch, tok, sp_: ("packed", [e["id"] for e in batch]),
        collate=lambda batch, tokenizer=None: ("padded", [e["id"] for e in batch]),
        write_micro_batch_slices=lambda tid, writer, mb, sp_: rec.write(tid, mb[0], mb[1]),
    )
    out = []
    next_id = 0
    for lens in pushes:
        for n in lens:
            ns["processed_entries_queue"].append({"id": next_id, "input_ids": [0] * n})
            next_id += 1
        # the main loop re-enters the block with batch_done = False until nothing more happens
        while True:
            before = len(rec.trace)
            ns["batch_done"] = False
            try:
                exec(code, ns)
            except IndexError:  # unpacked mode pops from an empty deque when short of samples
                break
            if len(rec.trace) == before and not ns["batch_done"]:
                break
            if not ns["processed_entries_queue"]:
                break
        out.append({"emitted": rec.trace, "published_samples": ns["published_samples"], "trainer_id": ns["trainer_id"]})
        rec.trace = []
    return out


def scenarios():
    rng = np.random.default_rng(0)
    sc = []
    for i, (nt, sp, tbs, gap, L, packing) in enumerate([
        (1, 1, 1, 8, 64, True), (2, 1, 1, 8, 64, True), (4, 1, 2, 8, 128, True), (4, 2, 1, 8, 96, True),
        (8, 1, 1, 16, 64, True), (2, 1, 4, 4, 64, False), (3, 1, 1, 6, 50, True), (4, 1, 1, 4, 32, True),
    ]):
        n_samples = 5 * tbs * gap
        if packing:
            lens = rng.integers(1, L + 1, size=n_samples).tolist()
        else:
            lens = rng.integers(1, L + 1, size=n_samples).tolist()
        # irregular arrival pattern
        pushes, k = [], 0
        while k < n_samples:
            m = int(rng.integers(1, 9))
            pushes.append(lens[k:k + m])
            k += m
        if not packing:  # unpacked mode needs whole batches available
            pushes = [lens[j:j + tbs * 2] for j in range(0, n_samples, tbs * 2)]
        sc.append({"name": f"s{i}", "params": dict(num_trainers=nt, seq_parallel=sp, train_batch_size=tbs,
                                                    gradient_accumulation_passes=gap, seq_length=L, seq_packing=packing),
                   "pushes": pushes})
    # resume from a non-zero published count
    sc.append({"name": "resume", "params": dict(num_trainers=2, seq_parallel=1, train_batch_size=1, gradient_accumulation_passes=4,
                                                seq_length=40, seq_packing=True, published_samples=12),
               "pushes": [rng.integers(1, 41, size=5).tolist() for _ in range(6)]})
    return sc


def main():
    out = []
    for s in scenarios():
        s["expected"] = run_reference(s["params"], s["pushes"])
        n = sum(len(e["emitted"]) for e in s["expected"])
        print(s["name"], "micro-batches:", n, "published:", s["expected"][-1]["published_samples"])
        out.append(s)
    (HERE / "schedule.json").write_text(json.dumps(out))




def reference_function_source(name: str) -> str:
    """Source text of a top-level function of the reference's preprocess.py."""
    lines = SRC.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(f"def {name}("))
    end = start + 1
    while end < len(lines) and (not lines[end].strip() or lines[end].startswith((" ", "\t"))):
        end += 1
    return "\n".join(lines[start:end])


def gen_filter_golden():
    """filter_zero_advantage_groups (preprocess.py:316-353) and _check_group_sizes (:70-83), executed
    from the reference source on small hand-made datasets."""
    ns = {"defaultdict": __import__("collections").defaultdict, "logger": types.SimpleNamespace(error=lambda *a, **k: None)}
    exec(compile(reference_function_source("filter_zero_advantage_groups"), "ref_filter", "exec"), ns)
    exec(compile(reference_function_source("_check_group_sizes"), "ref_check", "exec"), ns)
    rng = np.random.default_rng(1)
    cases = []
    for c in range(6):
        data = []
        for g in range(int(rng.integers(1, 5))):
            zero = bool(rng.integers(0, 2))
            for r in range(int(rng.integers(1, 4))):
                adv = 0.0 if zero else float(np.round(rng.normal(), 3))
                if not zero and r == 0 and rng.random() < 0.3:
                    adv = 5e-7  # below the epsilon
                data.append({"group_id": f"g{g}", "advantages": [adv] * int(rng.integers(1, 4)), "uid": len(data),
                             "metadata": {"rollout_index": r if rng.random() < 0.9 else 0}})
        order = rng.permutation(len(data)).tolist()
        data = [data[i] for i in order]
        kept, dropped = ns["filter_zero_advantage_groups"](list(data))
        sizes = {gs: bool(ns["_check_group_sizes"](data, gs)) for gs in (1, 2, 3)}
        cases.append({"data": data, "kept_uids": [e["uid"] for e in kept], "dropped": dropped, "group_size_ok": sizes})
    (HERE / "filter_groups.json").write_text(json.dumps(cases))
    print("filter_groups:", len(cases), "cases")


if __name__ == "__main__":
    main()
    gen_filter_golden()
