"""MicroBatchScheduler against golden traces produced by executing the reference's own scheduling
loop (tests/golden/make_schedule_golden.py), plus the invariants the trainer asserts on."""

import json

import pytest

from helpers import GOLDEN

SCENARIOS = json.loads((GOLDEN / "schedule.json").read_text())


def _run(params, pushes):
    from pipelinerl_amd.preprocess import MicroBatchScheduler

    sched = MicroBatchScheduler(length_of=lambda e: e["len"], **params)
    out = []
    next_id = 0
    for lens in pushes:
        sched.push({"id": next_id + k, "len": n} for k, n in enumerate(lens))
        next_id += len(lens)
        emitted = []
        while True:
            mbs, done = sched.drain()
            emitted += [[m.trainer_id, "sentinel" if m.sentinel else ("packed" if params["seq_packing"] else "padded"),
                         [s["id"] for s in m.samples]] for m in mbs]
            if (not mbs and not done) or not sched.queue:
                break
        out.append({"emitted": emitted, "published_samples": sched.published_samples, "trainer_id": sched.trainer_id})
    return out, sched


@pytest.mark.parametrize("scenario", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_schedule_matches_reference_trace(scenario):
    got, _ = _run(scenario["params"], scenario["pushes"])
    assert got == scenario["expected"]


@pytest.mark.parametrize("scenario", [s for s in SCENARIOS if s["params"]["seq_packing"]], ids=lambda s: s["name"])
def test_schedule_invariants(scenario):
    """What finetune_loop.py:674-675,859 relies on: within a step every lead trainer gets the same
    number of micro-batches (real + sentinel), quotas add up, token budgets are respected."""
    p = scenario["params"]
    lens = [n for push in scenario["pushes"] for n in push]
    got, sched = _run(p, scenario["pushes"])
    emitted = [e for step in got for e in step["emitted"]]
    leads = list(range(0, p["num_trainers"], p["seq_parallel"]))
    # round robin over lead trainers, in order
    assert [e[0] for e in emitted] == [leads[i % len(leads)] for i in range(len(emitted))]
    for tid, kind, ids in emitted:
        if kind != "sentinel":
            assert sum(lens[i] for i in ids) <= p["seq_length"]
    # samples are emitted in arrival order, each exactly once
    flat = [i for _, kind, ids in emitted for i in ids]
    assert flat == list(range(len(flat)))
    assert sched.published_samples - p.get("published_samples", 0) == len(flat)
    per_lead = sched.samples_per_lead_per_step
    start = p.get("published_samples", 0) // p["num_trainers"]
    for tid in leads:
        n = sum(len(ids) for t, _, ids in emitted if t == tid)
        assert n + start == sched.samples_per_trainer[tid]
        assert sched.samples_per_trainer[tid] <= sched.target_samples_per_lead
    assert per_lead * len(leads) == sched.samples_per_step
