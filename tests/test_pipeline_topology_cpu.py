"""BASELINE configs[2] / [3] TOPOLOGY - N learner ranks x M engines - executed end to end on host tensors (no GPU): six OS processes,
the product's stages / streams / schedulers / sample accounting / DDP group / weight-update group and protocol, with the three
device-only pieces replaced by tests/pipeline_cpu_hooks.py.  What must hold (reference preprocess.py:462-481,596-662,
finetune_loop.py:205-292,627-646,709,936-949, vllm1.py:64-127, world.py:192):

  * the preprocessor publishes to one partition per lead trainer, every rank consumes its per-step quota global_batch / N and the same
    number of micro-batches (sentinels fill), every rank takes every optimizer step;
  * trainer rank 0 + M engines form ONE weight-update group of M + 1 members; every engine receives version 0 and one update per
    optimizer step, acknowledges, and ends with the trainer's weights; `WeightUpdateSuccess` paces the actor;
  * the 2-learner run equals the 1-learner run on the same rollouts: the summed loss of step 0 is the same number (global normaliser:
    partial losses add), and the SGD update is the same direction at 1 / N of the size (DDP AVERAGES the ranks' gradients, as the reference's
    accelerate / DeepSpeed engines do; the loss is normalised by the GLOBAL samples_per_step on every rank, finetune_loop.py:644-646).
"""

import json
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
HERE = Path(__file__).resolve().parent


def _run(tmp_path, tag, n_learners, n_engines, steps=3, **kw):
    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    exp, cap = tmp_path / tag / "exp", tmp_path / tag / "cap"
    spec = PipelineSpec(exp_path=str(exp), model="tiny", global_batch=16, seq_length=48, attempts=4, steps=steps, optimizer="sgd", lr=0.05, param_dtype="fp32",
                        capture_step0=str(cap), n_problems=5, concurrent_groups=2, stage_timeout_s=240.0, learner="dropin", platform="cpu",
                        hooks="pipeline_cpu_hooks", sys_path=[str(HERE), str(ROOT)], n_learners=n_learners, n_engines=n_engines, weight_transport="gloo",
                        share_device=True, extra={"bucket_bytes": 4096}, **kw)
    res = run_pipeline(spec)
    assert "error" not in res, json.dumps(res.get("error"), indent=1)[:6000]
    return spec, res, cap


def test_two_learners_two_engines_over_gloo_match_one_learner(tmp_path):
    if str(HERE) not in sys.path:
        sys.path.insert(0, str(HERE))
    spec2, res2, cap2 = _run(tmp_path, "n2m2", 2, 2)
    st, s = res2["stages"], res2["summary"]
    bs, steps = spec2.global_batch, spec2.steps
    assert set(st) == {"engine0", "engine1", "learner0", "learner1", "preprocessor", "actor"}
    topo = s["topology"]
    assert topo["learners"] == 2 and topo["engines"] == 2 and topo["grad_backend"] == "gloo" and topo["weight_transport"] == "gloo_host_staged"
    # -- learners: lock-step, per-rank quota, same number of passes ----------------------------------------------------
    l0, l1 = st["learner0"], st["learner1"]
    assert l0["completed_steps"] == l1["completed_steps"] == steps and l0["samples"] == l1["samples"] == steps * bs
    assert l0["local_samples"] == l1["local_samples"] == steps * bs // 2, "each lead trainer consumed its quota"
    assert l0["micro_batches"] == l1["micro_batches"], "sentinels keep the ranks at the same number of forward/backward passes"
    assert (l0["rank"], l1["rank"], l0["world"]) == (0, 1, 2)
    # the schedule the preprocessor emitted: round robin over the two partitions, quota bs / 2 per step and trainer
    sched = st["preprocessor"]["schedule"]
    assert [t for t, _ in sched] == [k % 2 for k in range(len(sched))]
    assert {int(k): v for k, v in st["preprocessor"]["samples_per_trainer"].items()} == {0: steps * bs // 2, 1: steps * bs // 2}
    # -- weight-update group of M + 1: every engine got version 0 + one update per step and holds the trainer's weights ------
    for e in ("engine0", "engine1"):
        assert st[e]["updates"] == steps + 1 and st[e]["last_version"] == steps * bs
        assert st[e]["weight_group"]["size"] == 3 and st[e]["weight_group"]["bytes_received"] > 0
    assert {st["engine0"]["weight_group"]["rank"], st["engine1"]["weight_group"]["rank"]} == {1, 2}
    assert l0["weight_group"]["size"] == 3 and l1["weight_group"] is None, "only trainer rank 0 is in the weight-update group"
    assert l0["weight_group"]["bytes_sent"] == st["engine0"]["weight_group"]["bytes_received"] == st["engine1"]["weight_group"]["bytes_received"]
    assert s["engine_weights_equal_trainer_at_last_version"] is True and st["engine0"]["param_probe"] == st["engine1"]["param_probe"]
    assert len(l0["weight_sync"]["under_load_ms"]) == steps and l0["weight_sync"]["engines"] == 2
    # the actor saw the propagated versions (it is paced by them); it holds one llm per engine (the least busy one gets the next group,
    # actor.py:247-262 - with the synchronous scripted policy that is always the first)
    versions = {int(k) for k in st["actor"]["groups_per_model_version"]}
    assert versions <= {k * bs for k in range(steps + 1)} and len(versions) >= 2
    assert len(st["actor"]["llm_calls_per_engine"]) == 2 and sum(st["actor"]["llm_calls_per_engine"]) == st["actor"]["llm_calls"] > 0
    # replicas identical after the run (DDP): both ranks saved the same parameters after step 0
    a0, a1 = torch.load(cap2 / "rank0" / "params_after.pt"), torch.load(cap2 / "rank1" / "params_after.pt")
    assert all(torch.equal(a0[n], a1[n]) for n in a0)

    # -- the same rollouts through ONE learner and ONE engine ----------------------------------------------------------
    spec1, res1, cap1 = _run(tmp_path, "n1m1", 1, 1)
    assert set(res1["stages"]) == {"engine", "learner", "preprocessor", "actor"} and "topology" not in res1["summary"]
    b1 = torch.load(cap1 / "params_before.pt")
    b2 = torch.load(cap2 / "rank0" / "params_before.pt")
    assert all(torch.equal(b1[n], b2[n]) for n in b1), "the policy is a function of its seed"
    # step 0 covers the same samples: the first `bs` of the stream, split over the ranks
    def step0_lengths(captured):
        return sorted(int(b["seq_boundaries"][i + 1] - b["seq_boundaries"][i]) for b in captured if not b["sentinel"] for i in range(len(b["seq_boundaries"]) - 1))
    one = step0_lengths(torch.load(cap1 / "step0_batches.pt"))
    two = step0_lengths(torch.load(cap2 / "rank0" / "step0_batches.pt") + torch.load(cap2 / "rank1" / "step0_batches.pt"))
    assert one == two and len(one) == bs
    m1 = json.loads((cap1 / "step0_metrics.json").read_text())
    m2 = json.loads((cap2 / "rank0" / "step0_metrics.json").read_text())
    assert m2["rl/loss"] == pytest.approx(m1["rl/loss"], rel=1e-5, abs=1e-8), "partial losses of the ranks add up to the single-learner loss"
    assert m2["rl/num_output_tokens_sum"] == m1["rl/num_output_tokens_sum"]
    after1 = torch.load(cap1 / "params_after.pt")
    moved = 0
    for n in b1:
        d1, d2 = (after1[n] - b1[n]).double(), (a0[n] - b2[n]).double()
        if float(d1.abs().max()) == 0.0:
            continue
        moved += 1
        scale = float(d1.abs().max())
        assert float((d2 - 0.5 * d1).abs().max()) <= 2e-4 * scale + 1e-9, f"{n}: the 2-rank update is the 1-rank update / 2 (gradients averaged over ranks)"
    assert moved >= 3


def test_four_learners_three_engines_topology(tmp_path):
    """Odd engine count, four partitions: quotas of bs / 4, a weight-update group of 4."""
    if str(HERE) not in sys.path:
        sys.path.insert(0, str(HERE))
    spec, res, _ = _run(tmp_path, "n4m3", 4, 3, steps=2)
    st, s = res["stages"], res["summary"]
    bs = spec.global_batch
    assert s["topology"]["learners"] == 4 and s["topology"]["engines"] == 3
    assert len({st[f"learner{r}"]["micro_batches"] for r in range(4)}) == 1
    assert all(st[f"learner{r}"]["local_samples"] == 2 * bs // 4 for r in range(4))
    assert all(st[f"engine{e}"]["updates"] == 3 and st[f"engine{e}"]["weight_group"]["size"] == 4 for e in range(3))
    assert sorted(st[f"engine{e}"]["weight_group"]["rank"] for e in range(3)) == [1, 2, 3]
    assert s["engine_weights_equal_trainer_at_last_version"] is True


def test_tensor_parallel_engines_receive_only_their_slices(tmp_path):
    """BASELINE configs[4]'s receiver layout: two TP = 2 engines (four inference workers) + the trainer form one weight-update group PER TP
    RANK (vllm1.py:71 rank layout); every worker receives the slices `tp_shard.plan_tp_shards` assigns to its rank - about half of the
    parameter bytes - into vLLM-style stacked storage, and ends with exactly the trainer's slices."""
    if str(HERE) not in sys.path:
        sys.path.insert(0, str(HERE))
    spec, res, _ = _run(tmp_path, "n2m2tp2", 2, 2, steps=2, engine_tp=2)
    st, s = res["stages"], res["summary"]
    assert s["topology"]["engine_tp"] == 2 and s["engine_weights_equal_trainer_at_last_version"] is True
    l0 = st["learner0"]
    sent = l0["weight_group"]["bytes_sent_per_tp_rank"]
    assert l0["weight_group"]["size"] == 3 and len(sent) == 2  # each TP group: the trainer + that TP rank of both engines
    total = l0["weight_group"]["param_bytes"] * (spec.steps + 1)
    assert all(0.45 * total < b < 0.62 * total for b in sent), (sent, total)  # half of the sharded tensors + the replicated norms, per update
    ranks = sorted(r for e in ("engine0", "engine1") for r in st[e]["weight_group"]["ranks"])
    assert ranks == [1, 2, 3, 4]
    for e in ("engine0", "engine1"):
        assert st[e]["engine_tp"] == 2 and st[e]["updates"] == spec.steps + 1
        assert st[e]["weight_group"]["bytes_received_per_tp_rank"] == sent
        assert len(st[e]["param_probe"]) == 2 and st[e]["param_probe"][0] != st[e]["param_probe"][1]  # two ranks, two different halves
    assert st["engine0"]["param_probe"] == st["engine1"]["param_probe"] == l0["param_probes_per_tp_rank"][str(spec.steps * spec.global_batch)]


def test_spec_refuses_impossible_topologies():
    from pipelinerl_amd.pipeline_run import PipelineSpec

    with pytest.raises(ValueError, match="equal per-learner quotas"):
        PipelineSpec(exp_path="x", global_batch=10, n_learners=4)
    with pytest.raises(ValueError, match="share_device=True"):
        PipelineSpec(exp_path="x", weight_transport="ipc", share_device=False)
    with pytest.raises(ValueError, match="one GPU per member"):
        PipelineSpec(exp_path="x", weight_transport="rccl", share_device=True)
    with pytest.raises(ValueError, match="hooks"):
        PipelineSpec(exp_path="x", platform="cpu", weight_transport="gloo")
    # configs[2] and [3] as specs: engines first, learners after them (world.py:143-192)
    s = PipelineSpec(exp_path="x", model="7b", global_batch=4096, seq_length=8192, n_learners=2, n_engines=2, weight_transport="rccl", share_device=False)
    assert [str(s.device_of("engine", e)) for e in range(2)] == ["cuda:0", "cuda:1"] and [str(s.device_of("learner", r)) for r in range(2)] == ["cuda:2", "cuda:3"]
    assert s.learner_backend == "nccl" and s.stage_names() == ["engine0", "engine1", "learner0", "learner1", "preprocessor", "actor"]
    s = PipelineSpec(exp_path="x", model="7b", global_batch=4096, seq_length=8192, n_learners=4, n_engines=4, weight_transport="rccl", share_device=False)
    assert str(s.device_of("learner", 3)) == "cuda:7" and str(s.device_of("preprocessor")) == "cuda:4"


def test_baseline_configs_as_specs():
    """BASELINE.json configs[1..3] through `baseline_spec`: the GPU split is the reference's arithmetic (world.py:143-192, fractions 4 : 0 : 4)."""
    from pipelinerl_amd.pipeline_run import baseline_spec

    c1, c2, c3 = (baseline_spec(k, "x") for k in (1, 2, 3))
    assert (c1.model, c1.global_batch, c1.seq_length, c1.n_learners, c1.n_engines, c1.share_device, c1.weight_transport) == ("0p5b", 512, 2048, 1, 1, True, "ipc")
    assert (c2.model, c2.global_batch, c2.seq_length, c2.n_learners, c2.n_engines, c2.share_device, c2.weight_transport) == ("7b", 4096, 8192, 2, 2, False, "rccl")
    assert (c3.n_learners, c3.n_engines) == (4, 4) and c3.learner_backend == "nccl"
    assert [str(c3.device_of("engine", e)) for e in range(4)] == [f"cuda:{e}" for e in range(4)]
    assert [str(c3.device_of("learner", r)) for r in range(4)] == [f"cuda:{4 + r}" for r in range(4)]
    one_gpu = baseline_spec(2, "x", global_batch=16, share_device=True, weight_transport="ipc")  # the topology on one GPU, reduced batch
    assert one_gpu.n_learners == 2 and one_gpu.learner_backend == "gloo" and str(one_gpu.device_of("learner", 1)) == "cuda:0"
    c4 = baseline_spec(4, "x")  # Qwen2.5-32B: two TP = 2 engines on GPUs 0-3, four learners on 4-7, KL on, weight-update group of 5
    assert (c4.model, c4.n_learners, c4.n_engines, c4.engine_tp, c4.kl_coef, c4.weight_group_size) == ("32b", 4, 2, 2, 0.001, 5)
    assert [str(c4.device_of("engine", e, t)) for e in range(2) for t in range(2)] == ["cuda:0", "cuda:1", "cuda:2", "cuda:3"]
    assert str(c4.device_of("learner", 0)) == "cuda:4" and c4.shape["kv"] == 8
    with pytest.raises(ValueError, match="configs"):
        baseline_spec(5, "x")
    with pytest.raises(ValueError, match="engine_tp"):
        baseline_spec(1, "x", engine_tp=2)
