"""Host logic of the pipelined run (pipelinerl_amd/pipeline_run.py) that needs no GPU: the synthetic rollout plugin, the
training actor's pacing rule (reference actor.py:510-557), the engine-side in-flight update manager and its HTTP route
(vllm1.py:137-186, 244-249), the deferred-statistics learner step's accounting (finetune_loop.py:627-646, 698-713)."""

import json
import math
import threading
import time
import types
from pathlib import Path

import numpy as np
import pytest
import torch


# ---------------------------------------------------------------------------------------------
# synthetic plugin
# ---------------------------------------------------------------------------------------------
def test_synthetic_plugin_is_a_pure_function_of_its_configuration():
    import asyncio

    from pipelinerl_amd.synthetic_plugin import SyntheticLLM, generate_rollout, load_problems, synthetic_rollout

    problems = load_problems(["synthetic"], n_problems=3, seed=7)
    assert [p["id"] for p in problems] == [0, 1, 2] and all(p["seed"] == 7 for p in problems)
    a, b = SyntheticLLM(1000, 256), SyntheticLLM(1000, 256)
    for p in problems:
        for attempt in range(3):
            ra = asyncio.run(generate_rollout({}, a, p, None))
            rb = asyncio.run(generate_rollout({}, b, p, None))
            ta, tb = ra.training_texts[0], rb.training_texts[0]
            ta.check_consistency()
            assert ta.input_ids == tb.input_ids and ta.logprobs == tb.logprobs and ta.reward == tb.reward
            want = synthetic_rollout(7, p["id"], attempt, 1000, 256)
            assert ta.input_ids == want["input_ids"].tolist() and ta.prompt_tokens == want["prompt_len"]
            assert len(ta.input_ids) <= 256 and ta.labels[: ta.prompt_tokens] == [-100] * ta.prompt_tokens
            assert ta.finished == (len(ta.input_ids) < 256) and (not ta.finished or ta.input_ids[-1] == 2)
    # attempts of one problem differ, epochs differ
    x = synthetic_rollout(7, 0, 0, 1000, 256)["input_ids"]
    assert not np.array_equal(x[:8], synthetic_rollout(7, 0, 1, 1000, 256)["input_ids"][:8])
    assert not np.array_equal(x[:8], synthetic_rollout(7, 0, 0, 1000, 256, epoch=1)["input_ids"][:8])
    dense = synthetic_rollout(7, 0, 0, 1000, 256, dense=True)
    assert len(dense["input_ids"]) == 256 and not dense["finished"]


# ---------------------------------------------------------------------------------------------
# pacing
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("attempts,tbs,gap,wui,max_lag", [(8, 1, 512, 1, 512), (8, 1, 4096, 1, 1024), (4, 2, 8, 32, 10), (8, 1, 60, 1, 0), (3, 1, 10, 25, 7)])
def test_submission_budget_is_the_references_arithmetic(attempts, tbs, gap, wui, max_lag):
    """actor.py:510-534, restated term by term."""
    from pipelinerl_amd.actor_harness import ActorHarness

    total_batch_size = tbs * gap
    total_update_size = math.ceil(wui / total_batch_size) * total_batch_size
    groups_per_update = math.ceil(total_update_size / attempts)
    lag_groups = math.ceil(max_lag / attempts)
    assert ActorHarness.submission_budget(attempts, tbs, gap, wui, max_lag) == (lag_groups + groups_per_update, groups_per_update)
    assert ActorHarness.submission_budget(attempts, tbs, gap, wui, None) == (math.inf, None)


def test_paced_actor_never_runs_ahead_of_its_budget_and_stops_with_the_trainer(tmp_path):
    from pipelinerl_amd import streams
    from pipelinerl_amd.actor_harness import ActorHarness
    from pipelinerl_amd.synthetic_plugin import SyntheticLLM

    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        cfg = {"attempts": 4, "actor": {"rollout_policy": "pipelinerl_amd.synthetic_plugin.generate_rollout"},
               "dataset_loader": "pipelinerl_amd.synthetic_plugin.load_problems", "dataset_loader_params": {"n_problems": 3, "seed": 1},
               "train_dataset_names": ["synthetic"]}
        state = types.SimpleNamespace(propagated_weight_version=None, samples_processed=None)
        h = ActorHarness(cfg, [SyntheticLLM(100, 32, prompt_min=2, prompt_max=8)], tmp_path, trainer_state=state, wire="jsonl", shuffle_seed=0)
        bs, lag = 16, 8  # a step = 16 samples = 4 groups; the lag allows 2 more groups
        budget0, per_update = ActorHarness.submission_budget(4, 1, bs, 1, lag)
        assert (budget0, per_update) == (6, 4)
        seen = []

        def trainer():
            time.sleep(0.05)
            state.propagated_weight_version, state.samples_processed = 0, 0  # the actor waits for the first version
            for k in range(1, 4):
                # wait until the actor has used its whole budget, check that it does NOT go beyond it, then "train a step"
                want = budget0 + (k - 1) * per_update
                t0 = time.time()
                while h.published_groups < want and time.time() - t0 < 10:
                    time.sleep(0.002)
                time.sleep(0.05)
                seen.append((h.published_groups, want))
                state.samples_processed = k * bs
                state.propagated_weight_version = k * bs

        t = threading.Thread(target=trainer)
        t.start()
        n = h.run_paced(samples_target=3 * bs, train_batch_size=1, gradient_accumulation_passes=bs, weight_update_interval=1, max_lag=lag,
                        concurrent_groups=3, poll_s=0.001)
        t.join()
        assert seen == [(6, 6), (10, 10), (14, 14)], seen
        assert n == h.published_samples and h.timing["blocked_by_lag_s"] > 0 and h.timing["versions_seen"] >= 3
        # every record is one group; rollouts are stamped with the version that had propagated when they started
        with streams.read_stream(streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")) as r:
            records = []
            for rec in r.read():
                records.append(rec)
                if len(records) == h.published_groups:
                    break
        assert all(len(g) == 4 and len({e["group_id"] for e in g}) == 1 for g in records)
        versions = [g[0]["metadata"]["model_version"] for g in records]
        assert versions[:6] == [0] * 6 and versions == sorted(versions) and set(versions) <= {0, 16, 32, 48}
        # problems repeat epoch after epoch with fresh draws
        assert records[0][0]["input_ids"] != records[3][0]["input_ids"]
    finally:
        streams.reset_streams_backend()


# ---------------------------------------------------------------------------------------------
# engine-side update manager + HTTP route
# ---------------------------------------------------------------------------------------------
class _Worker:
    def __init__(self, log, delay=0.0):
        self.log, self.delay, self.received = log, delay, []

    def receive_weight_update(self, request_json):
        self.log.append(("update_begin", json.loads(request_json)["version"]))
        time.sleep(self.delay)
        self.received.append(json.loads(request_json))
        self.log.append(("update_end", json.loads(request_json)["version"]))

    def init_actor_update_group(self, *a):
        self.log.append(("init", a))

    def close_communicator(self):
        self.log.append(("close",))


def test_inflight_update_pauses_updates_resumes_and_serialises(tmp_path):
    from pipelinerl_amd.engine_update import InflightUpdateManager, ScriptedEngine, UpdateServer
    from pipelinerl_amd.finetune_loop import ParameterInfo, WeightUpdateRequest, _http_post

    log = []
    quanta_during_update = []

    def generate_step():
        time.sleep(0.002)
        if any(e[0] == "update_begin" for e in log[-1:]):
            quanta_during_update.append(1)

    workers = [_Worker(log, delay=0.05), _Worker(log, delay=0.0)]
    engine = ScriptedEngine(workers, generate_step)
    manager = InflightUpdateManager(engine)
    server = UpdateServer(manager)
    try:
        time.sleep(0.05)
        before = engine.quanta
        assert before > 0, "the scripted engine generates while nothing else happens"
        msgs = [WeightUpdateRequest(version=v, parameters_info=[ParameterInfo(name="w", shape=[2], dtype="torch.bfloat16")], transport="ipc") for v in (8, 16)]
        # two trainers posting at once: the manager's lock runs the updates one after the other (vllm1.py:160)
        ts = [threading.Thread(target=_http_post, args=(server.url + "/receive_weight_update", m.model_dump())) for m in msgs]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        kinds = [e[0] for e in log]
        assert kinds.count("update_begin") == 4 and kinds.count("update_end") == 4
        # per version: both workers begin and end before the other version starts
        order = [e[1] for e in log if e[0] == "update_begin"]
        assert order in ([8, 8, 16, 16], [16, 16, 8, 8])
        assert not quanta_during_update, "generation ran while an update was being applied"
        assert [w.received[0]["version"] for w in workers] == [order[0]] * 2 and workers[0].received[0]["transport"] == "ipc"
        assert len(manager.timings) == 2 and all(t["update_s"] >= 0.05 and t["pause_s"] >= 0 and t["resume_s"] >= 0 for t in manager.timings)
        time.sleep(0.05)
        assert engine.quanta > before, "generation resumed after the updates (mode keep: nothing was dropped)"
        assert engine.current_version == order[-1] and sum(engine.quanta_by_version.values()) == engine.quanta
        # an unknown route is refused, a failing worker answers 500 and the engine still resumes
        import requests

        assert requests.post(server.url + "/nope", json={}).status_code == 404
        workers[0].receive_weight_update = lambda _r: (_ for _ in ()).throw(ValueError("model x not found in model state dict"))
        r = requests.post(server.url + "/receive_weight_update", json=msgs[0].model_dump())
        assert r.status_code == 500 and "not found" in r.json()["error"]
        q = engine.quanta
        time.sleep(0.05)
        assert engine.quanta > q
    finally:
        engine.shutdown()
        server.close()


def test_pause_right_after_resume_still_waits_for_the_quantum():
    """Back-to-back updates (a second one queued on the manager's lock pauses immediately after the first one's resume): `pause_generation`
    must not return on the strength of the PREVIOUS park while the loop thread is already past its check - no generation quantum may
    overlap the window between pause and resume."""
    import asyncio

    from pipelinerl_amd.engine_update import ScriptedEngine

    copying = threading.Event()
    overlaps = []

    def generate_step():
        if copying.is_set():
            overlaps.append("started inside the window")
        time.sleep(0.0002)
        if copying.is_set():
            overlaps.append("ran into the window")

    engine = ScriptedEngine([], generate_step)

    async def hammer():
        for _ in range(400):
            await engine.pause_generation()
            assert not engine.generating()
            copying.set()
            await asyncio.sleep(0.0003)   # the weight copy
            copying.clear()
            await engine.resume_generation()  # ... and the next update pauses right away

    try:
        asyncio.run(hammer())
        assert not overlaps, overlaps[:3]
        q = engine.quanta
        time.sleep(0.05)
        assert engine.quanta > q  # left running
    finally:
        engine.shutdown()


def test_pause_mode_other_than_keep_is_refused():
    import asyncio

    from pipelinerl_amd.engine_update import ScriptedEngine

    e = ScriptedEngine([], None)
    with pytest.raises(ValueError):
        asyncio.run(e.pause_generation(mode="abort"))
    asyncio.run(e.pause_generation())
    asyncio.run(e.resume_generation())
    e.shutdown()


# ---------------------------------------------------------------------------------------------
# StreamedLearnerStep: accounting with the statistics deferred to the step boundary
# ---------------------------------------------------------------------------------------------
class _FakeFusedModel(torch.nn.Module):
    """Stands in for a model prepared with install_fused_head: `model(rl_batch=...)` -> (loss, device statistics)."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))
        self._prl_fused_head = {"chunk_rows": 8192}
        self.calls = 0

    def forward(self, rl_batch=None, rl_config=None, current_step=0, max_step=1):
        from pipelinerl_amd import _lib

        self.calls += 1
        stats = torch.zeros(_lib.PRL_NUM_STATS, dtype=torch.float64)
        n_seq = len(rl_batch.seq_boundaries) - 1
        labelled = int((rl_batch.labels != -100).sum())
        loss = self.w.sum() * float(labelled)
        stats[_lib.STAT_INDEX["loss"]] = float(labelled)
        stats[_lib.STAT_INDEX["num_sequences"]] = n_seq
        stats[_lib.STAT_INDEX["num_output_tokens_sum"]] = labelled
        stats[_lib.STAT_INDEX["reward"]] = 0.5 * n_seq
        if rl_batch.model_extra.get("poison"):
            stats[_lib.STAT_INDEX["nonfinite_new_logprobs"]] = 1
        return loss, stats


def _packed(n_seq, length, version=0):
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    T = n_seq * length
    ids = torch.arange(T)[None] % 50 + 3
    labels = ids.clone()
    labels[:, ::length] = -100
    f = torch.zeros(1, T)
    return PipelineBatchEncoding(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), position_ids=(torch.arange(T) % length)[None],
                                 segment_ids=(torch.arange(T) // length)[None], rewards=f, advantages=f, ref_logprobs=f, old_logprobs=f,
                                 group_tokens=f + 1, num_labels=f + 1, overflow=f, model_version=version, is_packed=True,
                                 seq_boundaries=torch.arange(0, T + 1, length, dtype=torch.int32))


def test_streamed_learner_step_defers_statistics_to_the_boundary(tmp_path):
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.utils import create_sentinel_batch
    from pipelinerl_amd.finetune_loop import TRAINER_TOPIC, LearnerStep, StreamedLearnerStep, annotate_host_batch
    from pipelinerl_amd.state import TrainerState

    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        model = _FakeFusedModel()
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        sent = []
        mgr = types.SimpleNamespace(send_weight_update=lambda v: sent.append(v))
        step = StreamedLearnerStep(model, opt, RLConfig(), train_batch_size=1, gradient_accumulation_passes=6, max_train_steps=4,
                                   weight_update_manager=mgr, weight_update_interval=1,
                                   trainer_stream=streams.SingleStreamSpec(exp_path=tmp_path, topic=TRAINER_TOPIC))
        # step 1: micro-batches of 2 + 3 + 1 sequences; statistics arrive only with the optimizer step
        b1, b2, b3 = (annotate_host_batch(_packed(n, 5)) for n in (2, 3, 1))
        assert b1.model_extra["tokens"] == 10 and b1.model_extra["labelled_rows"].tolist() == [0, 1, 2, 3, 5, 6, 7, 8]
        r = step.step(b1)
        assert r["stats"] is None and not r["did_optimizer_step"] and step.total_samples == 2
        r = step.step(b2)
        assert r["stats"] is None and step.total_samples == 5
        r = step.step(b3)
        assert r["did_optimizer_step"] and r["stats"]["num_output_tokens_sum"] == 4
        assert r["metrics"]["rl/loss"] == 8 + 12 + 4 and r["metrics"]["rl/reward"] == pytest.approx(0.5)
        assert step.metrics.samples == 6 and step.metrics.completed_steps == 1 and step.metrics.tokens == 30 and step.metrics.passes == 3
        assert step.maybe_send_weights() and sent == [6] and not step.maybe_send_weights()
        # step 2 on rollouts of version 0 and 6: the lag of a micro-batch = samples trained when it is consumed - its model version
        step.step(annotate_host_batch(_packed(5, 4, version=0)))
        r = step.step(annotate_host_batch(_packed(1, 4, version=6)))
        assert r["did_optimizer_step"] and step.metrics.samples == 12
        assert list(step.lag_samples) == [0, 0, 0, 6, 0] and step.lag_samples.maxlen == 4096
        # a non-finite value is reported at the END of its step, by the reference's assert text
        step2 = StreamedLearnerStep(_FakeFusedModel(), opt, RLConfig(), train_batch_size=1, gradient_accumulation_passes=2, max_train_steps=4)
        bad = annotate_host_batch(_packed(1, 5))
        bad.model_extra["poison"] = True
        assert step2.step(bad)["stats"] is None
        with pytest.raises(AssertionError, match="new_logprobs is not finite"):
            step2.step(annotate_host_batch(_packed(1, 5)))
        # a sentinel batch runs forward / backward, adds no statistics and no samples
        step3 = StreamedLearnerStep(_FakeFusedModel(), opt, RLConfig(), train_batch_size=1, gradient_accumulation_passes=2, max_train_steps=4)
        step3.step(annotate_host_batch(_packed(2, 5)))
        s = create_sentinel_batch(None, tokenizer=types.SimpleNamespace(eos_token_id=2), model_version=0)
        assert step3.metrics.completed_steps == 1
        n_before = step3.model.calls
        step3.step(s)
        assert step3.model.calls == n_before + 1 and step3.total_samples == 2 and not step3._stats_dev
        # a model without the fused head is refused
        with pytest.raises(TypeError):
            StreamedLearnerStep(torch.nn.Linear(2, 2), opt, RLConfig(), train_batch_size=1, gradient_accumulation_passes=2, max_train_steps=4)
        step.finish()
        st = TrainerState(tmp_path)
        st.start_listening()
        assert st.wait_for_training_done(timeout=10) and st.samples_processed == 12
        assert isinstance(step, LearnerStep)
    finally:
        streams.reset_streams_backend()


def test_pipeline_spec_defaults_are_baseline_config_1():
    from pipelinerl_amd.pipeline_run import MODEL_SHAPES, PipelineSpec, rl_config_of
    from pipelinerl_amd.weight_sync_probe import qwen25_shapes

    spec = PipelineSpec(exp_path="/tmp/x")
    assert (spec.model, spec.global_batch, spec.seq_length, spec.attempts, spec.chunk_n_groups, spec.weight_update_interval) == ("0p5b", 512, 2048, 8, 2, 1)
    assert spec.lag == 512 and spec.shape["vocab"] == 151936 and spec.shape["tied"]
    rl = rl_config_of(spec)
    assert (rl.policy_loss, rl.epsilon_low, rl.kl_coef, rl.batch_size, rl.divide_advantage_by_std) == ("ppo", 0.02, 0.0, 512, False)
    # the shapes agree with the weight-sync parameter sets
    for name in ("0p5b", "7b"):
        s = MODEL_SHAPES[name]
        shapes = dict(qwen25_shapes(name))
        assert shapes["model.embed_tokens.weight"] == (s["vocab"], s["hidden"]) and ("lm_head.weight" in shapes) == (not s["tied"])
        assert shapes["model.layers.0.mlp.gate_proj.weight"] == (s["inter"], s["hidden"])


def test_bench_refuses_to_run_n_gpus_as_one_rank():
    """`python bench.py --gpus N` on a node with fewer devices exits non-zero and prints no line (it used to continue as ONE rank and
    report n_gpus 1 with the whole batch on GPU 0); a launcher that started a different number of ranks is refused too."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 7
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PRL_BENCH_SHARE_DEVICE")}
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(n), "--workload", "tiny"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "HIP devices" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4", "--workload", "tiny"], capture_output=True, text=True, timeout=300,
                         env={**env, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_native_learner_step_recognises_an_actor_critic_model():
    """rl_step's value-head branch (rl/__init__.py:162, 265-272, 367-381) is part of NativeLearnerStep: a model with a `value_head`, bare or
    behind a `.module` wrapper, is detected at construction (its step then routes `outputs.value` through the value kernel -
    tests/test_gpu_value_head.py); a plain causal LM is not."""
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune_loop import NativeLearnerStep

    class ActorCritic(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pretrained_model = torch.nn.Linear(2, 2)
            self.value_head = torch.nn.Linear(2, 1)

    m = ActorCritic()
    kw = dict(eos_token_id=2, samples_per_step=8, max_train_steps=2)
    assert NativeLearnerStep(m, torch.optim.SGD(m.parameters(), lr=0.1), RLConfig(), **kw).has_value_head is True
    wrapped = types.SimpleNamespace(module=m, parameters=m.parameters)
    assert NativeLearnerStep(wrapped, torch.optim.SGD(m.parameters(), lr=0.1), RLConfig(), **kw).has_value_head is True
    plain = torch.nn.Linear(2, 2)
    assert NativeLearnerStep(plain, torch.optim.SGD(plain.parameters(), lr=0.1), RLConfig(), **kw).has_value_head is False


def test_summary_of_a_recorded_pipeline_run_recomputes():
    """`pipeline_run.summarize` over the stage reports of a run recorded on an MI355X (profiles/r05s_*: the 7B-shaped pipeline) gives the
    summary that run printed, and the summary is consistent with itself: throughput = batch / step time, the busy fractions are
    fractions, every micro-batch has a lag, the sum of the stages' busy seconds is what `overlap` quotes."""
    import json

    from pipelinerl_amd.pipeline_run import PipelineSpec, summarize

    rec = json.loads((Path(__file__).resolve().parent.parent / "profiles" / "r05s_pipeline_7b_shape_bs16_seq8192.json").read_text())
    spec = PipelineSpec(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in rec["spec"].items() if k in PipelineSpec.__dataclass_fields__})
    assert spec.model == "7b" and spec.global_batch == 16 and spec.lag == 16 and spec.budget == 8192
    s = summarize(spec, rec["stages"])
    assert json.loads(json.dumps(s)) == rec["summary"]
    assert s["samples_per_s"] == pytest.approx(spec.global_batch / s["s_per_step"], rel=1e-9)
    assert all(0.0 <= f <= 1.0 for f in s["busy_frac"].values()) and s["busy_frac"]["learner"] > 0.9
    assert sum(s["lag_optimizer_steps_histogram"].values()) == rec["stages"]["learner"]["micro_batches"]
    assert s["overlap"]["stages_back_to_back_s_per_step"] == pytest.approx(sum(s["stage_busy_s_per_step"].values()))
    assert s["weight_sync_under_load_ms"]["updates"] == s["optimizer_steps"] == 3 and s["engine_weights_equal_trainer_at_last_version"] is True


def test_step_granular_loop_names_where_gspo_lives():
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.hotpath import HotPathStep

    with pytest.raises(ValueError, match="rl_step_fused_head"):
        HotPathStep(RLConfig(policy_loss="gspo"), eos_token_id=2)
