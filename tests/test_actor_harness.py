"""The plugin surface is CALLED (round-2 review row J1): `cfg.actor.rollout_policy` / `cfg.dataset_loader` resolved by
dotted path, `attempts` rollouts per problem on a scripted llm, group stamping, one `actor` record per group -
reference actor.py:141, 176-225, 648-652, 803-808.  The GPU half of the flow (records -> PreprocessorLoop ->
LearnerStep) is tests/test_gpu_actor_flow.py."""

import asyncio
import json
import os
import sys
import types
from pathlib import Path

import pytest

from helpers import GOLDEN

CFG = {
    "attempts": 4,
    "actor": {"rollout_policy": "plugins.guessing_plugin.generate_guessing_rollout", "rollout_retry_initial_delay_s": 0.001},
    "dataset_loader": "plugins.guessing_plugin.load_problems",
    "train_dataset_names": ["train"], "test_dataset_names": ["test"],
    "train_subset": {"begin": 10, "end": 16},
}


@pytest.fixture()
def streams(tmp_path):
    from pipelinerl_amd import streams as s

    s.reset_streams_backend()
    yield s
    s.clean_shm_streams(tmp_path)
    s.reset_streams_backend()


def test_dataset_plugin_returns_what_the_reference_loader_returns():
    from plugins.guessing_plugin import load_problems

    g = json.loads((GOLDEN / "guessing_problems.json").read_text())
    for split in ("train", "test"):
        want = [{"answer": a, "dataset": split, "domain": g["domain"]} for a in g["answers"][split]]
        assert load_problems([split]) == want
    assert load_problems(["train", "test"]) == load_problems(["train"]) + load_problems(["test"])
    assert load_problems(["other"]) == []


@pytest.mark.skipif(not os.path.isdir("/root/reference/pipelinerl"), reason="the reference checkout is only present in the build container")
def test_harness_drives_the_references_own_dataset_plugin(streams, tmp_path):
    """`cfg.dataset_loader` pointing INTO the reference: its `load_problems` supplies the problems."""
    sys.path.insert(0, str(GOLDEN))
    try:
        from make_guessing_golden import import_reference_guessing

        ref = import_reference_guessing()
    finally:
        sys.path.remove(str(GOLDEN))
    sys.modules["reference_guessing"] = ref
    try:
        from pipelinerl_amd.actor_harness import ActorHarness
        from plugins.guessing_plugin import ScriptedLLM

        streams.set_streams_backend("files")
        h = ActorHarness({**CFG, "dataset_loader": "reference_guessing.load_problems", "train_subset": None}, [ScriptedLLM()], tmp_path)
        problems = h.load_problems()
        assert len(problems) == 512 and problems[1] == {"answer": 383, "dataset": "train", "domain": "guessing"}
        assert h.run(problems[:2]) > 0
    finally:
        del sys.modules["reference_guessing"]


@pytest.mark.parametrize("wire,backend", [("jsonl", "files"), ("ragged", "shm"), ("jsonl", "shm")])
def test_groups_are_rolled_out_stamped_and_published(streams, tmp_path, libprl, wire, backend):
    from pipelinerl_amd.actor_harness import ActorHarness
    from pipelinerl_amd.ragged import RaggedRollouts
    from pipelinerl_amd.rollouts import TrainingText
    from plugins.guessing_plugin import ScriptedLLM

    streams.set_streams_backend(backend, **({"mirror_jsonl": ["actor"]} if backend == "shm" else {}))
    state = types.SimpleNamespace(propagated_weight_version=7)
    llms = [ScriptedLLM(flaky_calls=(3, 17, 18)), ScriptedLLM(split=0.4)]
    h = ActorHarness(CFG, llms, tmp_path, trainer_state=state, scheduler_name="sched3", wire=wire, shuffle_seed=0)
    problems = h.load_problems()
    assert len(problems) == 6 and problems[0]["answer"] == (2 * 10 * 191) % 1024 + 1  # train_subset applied
    n = h.run(problems)
    assert n == h.published_samples and h.published_groups == 6
    assert h.retries == 3, "the flaky llm raised three retryable time-outs; each restarted its rollout"
    assert llms[0].calls > 0 and llms[1].calls > 0, "rollouts are spread over the llms (least busy first)"
    # read the stream back: one record per group
    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")
    groups = []
    with streams.read_stream(spec) as r:
        for rec in r.read():
            groups.append(rec.to_entries() if isinstance(rec, RaggedRollouts) else rec)
            if len(groups) == 6:
                break
    assert sum(len(g) for g in groups) == n
    rewards_seen = set()
    for gi, g in enumerate(groups):
        assert {e["group_id"] for e in g} == {f"sched3_{gi}"}
        assert {e["metadata"]["rollout_index"] for e in g} == {0, 1, 2, 3}  # `attempts` rollouts, shuffled
        assert all(e["metadata"]["model_version"] == 7 for e in g)
        by_rollout = {}
        for e in g:
            by_rollout.setdefault(e["metadata"]["rollout_index"], []).append(e["metadata"]["step_index"])
            if wire == "jsonl":
                TrainingText(**e).check_consistency()
        assert all(steps == list(range(len(steps))) for steps in by_rollout.values())  # one text per turn, in order
        # the interval search finds 1..1024 within 13 guesses: reward 2 - (turns - 1) / 10, the same on every text of a rollout
        assert all(e["reward"] == pytest.approx(2 - (len(by_rollout[e["metadata"]["rollout_index"]]) - 1) / 10) for e in g)
        assert RaggedRollouts.from_entries(g).n_seqs == len(g)  # directly ingestible by the preprocessor
        rewards_seen.update(e["reward"] for e in g)
    assert len(rewards_seen) > 1, "two differently scripted llms: the group baseline has something to subtract"
    if backend == "shm":  # the JSONL mirror holds the reference's text record: replayable with backend=files
        lines = (tmp_path / "streams" / "actor" / "0" / "0" / "0.jsonl").read_text().splitlines()
        assert len(lines) == 6 and len(json.loads(lines[0])) == len(groups[0])


def test_fatal_and_sync_policies(streams, tmp_path):
    from pipelinerl_amd.actor_harness import ActorHarness
    from plugins.guessing_plugin import ScriptedLLM, load_problems

    streams.set_streams_backend("files")
    # a policy that is a plain function is accepted (domains/dispatcher.py:84-86)
    h = ActorHarness({**CFG, "attempts": 1, "actor": {"rollout_policy": "plugins.guessing_plugin.sync_rollout"}}, [ScriptedLLM()], tmp_path)
    group = asyncio.run(h.rollout_group(load_problems(["test"])[0], 0))
    assert len(group) == 1 and group[0].metrics.success and group[0].dataset_name == "test"
    # malformed answers: the policy's own error branch (negative reward, no_answer), not an exception
    h = ActorHarness({**CFG, "attempts": 2}, [ScriptedLLM(malformed_after=2)], tmp_path)
    group = asyncio.run(h.rollout_group(load_problems(["train"])[3], 1))
    assert all(r.metrics.no_answer and r.training_texts[-1].reward == pytest.approx(-2 + 2 / 10) for r in group)
    # retry budget exhausted -> the exception surfaces (the reference stops all rollout tasks, actor.py:160-170)
    h = ActorHarness({**CFG, "actor": {**CFG["actor"], "max_rollout_retries": 1}}, [ScriptedLLM(always_fail=True)], tmp_path)
    with pytest.raises(TimeoutError):
        asyncio.run(h.rollout_group(load_problems(["train"])[0], 2))
    # a non-retryable exception is never retried
    class Broken:
        def generate(self, messages):
            raise KeyError("boom")
    h = ActorHarness(CFG, [Broken()], tmp_path)
    with pytest.raises(KeyError):
        asyncio.run(h.rollout_group(load_problems(["train"])[0], 3))
    assert h.retries == 0
    with pytest.raises(ValueError):
        ActorHarness(CFG, [ScriptedLLM()], tmp_path, wire="protobuf")
