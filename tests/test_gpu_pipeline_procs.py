"""The hot path as FOUR PROCESSES on one GPU (pipelinerl_amd/pipeline_run.py: actor harness -> shm streams ->
PreprocessorLoop -> StreamedLearnerStep with the fused head -> colocated weight hand-off through the engine-side update
manager), and a WHOLE optimizer step of it against the oracle driven over the same records:

  * the `actor` records the run produced (JSONL mirror, the reference's text record) go through `oracle.preprocess`
    (preprocess_fn + populate_rl_data, reference rl/__init__.py:453-594) chunk by chunk and through an independent
    restatement of the packing rule (preprocess.py:610-625); every micro-batch the learner consumed in step 0 must equal the
    oracle's `collate_packed` of the same samples - integers bit for bit, floats to 1e-6;
  * the policy is rebuilt from its seed; per micro-batch its fp32 logits (hidden states in bf16 times the fp32 head, the
    reference's `apply_fp32_lm_head`, checkpoints.py:87-103) go through `oracle.rl_loss_torch.rl_step`; the summed loss, the
    aggregated statistics (`aggregate_rl_stats`, finetune_loop.py:908-922) and the parameter delta after the SGD step (autograd
    through the same model, fed the oracle's d loss / d logits) are compared with what the pipelined learner reported and
    saved: 1e-4 relative for loss and statistics (north_star), 1e-4 relative (2-norm, per tensor) for the accumulated gradients, one ulp for the parameters after the step."""

import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _first_fit(lengths, seq_length, quota):
    """preprocess.py:610-625 for one trainer: pop while the next sample fits, flush on overflow or when the step's quota is full."""
    out, cur, used, taken = [], [], 0, 0
    for i, n in enumerate(lengths):
        if taken == quota:
            break
        if cur and used + n > seq_length:
            out.append(cur)
            cur, used = [], 0
        cur.append(i)
        used += n
        taken += 1
        if taken == quota:
            out.append(cur)
            cur = []
    return out


@pytest.mark.parametrize("learner", ["streamed", "dropin", "streamed+compact_wire"])
def test_four_process_pipeline_and_step0_vs_oracle(libprl, cuda_device, tmp_path, learner):
    learner, _, wire = learner.partition("+")
    wire = "compact" if wire else "full"  # compact: the pack kernel runs in the LEARNER's loader; step 0 must still be the oracle's
    from oracle import preprocess as opre
    from oracle import rl_loss_torch as orlt
    from pipelinerl_amd.finetune.rl.utils import aggregate_rl_stats
    from pipelinerl_amd.pipeline_run import PipelineSpec, build_policy, rl_config_of, run_pipeline

    exp, cap = tmp_path / "exp", tmp_path / "cap"
    bs, seq, lr = 16, 128, 0.05
    spec = PipelineSpec(exp_path=str(exp), model="tiny", global_batch=bs, seq_length=seq, attempts=4, steps=3, optimizer="sgd", lr=lr,
                        param_dtype="fp32", mirror_jsonl=True, capture_step0=str(cap), n_problems=5, concurrent_groups=2,
                        stage_timeout_s=600.0, learner=learner, engine_load=True, wire=wire)
    res = run_pipeline(spec)
    assert "error" not in res, json.dumps(res.get("error"), indent=1)[:4000]
    s, st = res["summary"], res["stages"]
    # -- the run as a pipeline -----------------------------------------------------------------------------
    assert s["optimizer_steps"] == 3 and st["learner"]["samples"] == 3 * bs
    assert st["engine"]["updates"] == 4 and st["engine"]["last_version"] == 3 * bs, "version 0 + one update per optimizer step"
    assert s["engine_weights_equal_trainer_at_last_version"] is True
    assert st["engine"]["generation_quanta"] > 0, "the scripted engine generated between the updates"
    assert st["actor"]["published_samples"] >= 3 * bs and st["preprocessor"]["published_samples"] >= 3 * bs
    budget0, per_update = st["actor"]["pacing"]["budget"]
    assert st["actor"]["published_groups"] <= budget0 + 3 * per_update, "the actor stayed inside its max_lag budget"
    versions = {int(k): v for k, v in st["actor"]["groups_per_model_version"].items()}
    assert set(versions) <= {0, bs, 2 * bs, 3 * bs} and versions.get(0, 0) >= budget0 - per_update
    assert len(st["learner"]["weight_sync"]["under_load_ms"]) == 3 and all(x > 0 for x in st["learner"]["weight_sync"]["under_load_ms"])
    lag = {int(k): v for k, v in s["lag_optimizer_steps_histogram"].items()}
    assert sum(lag.values()) == st["learner"]["micro_batches"] and min(lag) >= 0 and max(lag) <= 2
    assert all(0.0 <= f <= 1.0 for f in s["busy_frac"].values())

    # -- step 0 against the oracle over the same records ---------------------------------------------------
    groups = [json.loads(line) for line in (exp / "streams" / "actor" / "0" / "0" / "0.jsonl").read_text().splitlines()]
    samples = []
    for lo in range(0, len(groups), spec.chunk_n_groups):  # the preprocessor's chunks (preprocess.py:206-210)
        chunk = [e for g in groups[lo: lo + spec.chunk_n_groups] for e in g]
        samples += opre.preprocess_chunk(chunk, 2, False)
    plan = _first_fit([len(e["input_ids"]) for e in samples], seq, bs)
    captured = torch.load(cap / "step0_batches.pt")
    assert [len(b["seq_boundaries"]) - 1 for b in captured] == [len(p) for p in plan], "micro-batch composition of step 0"
    want_batches = [opre.collate_packed([samples[i] for i in p], 2, 1) for p in plan]
    for got, want in zip(captured, want_batches):
        for k, w in want.items():
            if isinstance(w, np.ndarray):
                g = got[k].numpy()
                if w.dtype.kind in "iu":
                    np.testing.assert_array_equal(g, w, err_msg=k)
                else:
                    np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-7, err_msg=k)
            else:
                assert got[k] == w, k

    model = build_policy(spec, cuda_device, seed=spec.seed)
    before = torch.load(cap / "params_before.pt")
    for n, p in model.named_parameters():
        assert torch.equal(p.detach().cpu(), before[n]), f"the policy is a pure function of its seed: {n}"
    cfg = rl_config_of(spec)
    total_loss, stats = 0.0, {}
    for want in want_batches:
        ids = torch.from_numpy(want["input_ids"]).to(cuda_device)
        hidden = model.model(input_ids=ids, attention_mask=torch.ones_like(ids), position_ids=torch.from_numpy(want["position_ids"]).to(cuda_device))[0]
        # the head reads the hidden states in bf16; the VALUE is rounded, the gradient is not (autograd's own backward of a bf16
        # cast would round d hidden to bf16 - right for the reference's bf16 models, 2^-9 too coarse for this fp32 check)
        hq = hidden + (hidden.to(torch.bfloat16).float() - hidden).detach()
        # the head's product in fp64: the reference side must not be the less accurate one (the fused head carries fp32-GEMM accuracy)
        logits = (hq.double() @ model.lm_head.weight.double().t()).float()
        out = orlt.rl_step(logits.detach().cpu().numpy(), want, cfg, 0, spec.steps, True)
        total_loss += float(out["loss"])
        for k, v in out["stats"].items():
            stats.setdefault(k, []).append(v)
        logits.backward(out["grad_logits"].to(cuda_device))
    metrics = json.loads((cap / "step0_metrics.json").read_text())
    assert metrics["rl/loss"] == pytest.approx(total_loss, rel=1e-4, abs=1e-7)
    want_metrics = aggregate_rl_stats(stats, bs)
    for k, w in want_metrics.items():
        assert metrics[k] == pytest.approx(w, rel=1e-4, abs=1e-6), k
    # the gradients the optimizer saw (1e-4: the head's products carry fp32-GEMM accuracy, the body is the same code on both sides) ...
    grads = torch.load(cap / "grads_step0.pt")
    after = torch.load(cap / "params_after.pt")
    worst = {}
    for n, p in model.named_parameters():
        want, got = p.grad.cpu().double(), grads[n].double()
        assert float(want.abs().max()) > 0, f"{n} got no gradient"
        worst[n] = float((got - want).norm() / want.norm())
        # ... and the parameters after the step: before - lr * gradient, to the rounding of an fp32 parameter (the delta is ~1e-5 of a
        # weight, so half an ulp of the weight is all that separates the two sides)
        want_after = (before[n].double() - lr * want).float()
        ulp = torch.finfo(torch.float32).eps * torch.maximum(before[n].abs(), want_after.abs())
        assert bool(((after[n] - want_after).abs() <= 1.01 * ulp + 1e-3 * lr * float(want.abs().max())).all()), f"parameters after step 0: {n}"
        assert not torch.equal(after[n], before[n]), f"{n} did not move"
    bad = {n: e for n, e in worst.items() if e > 1e-4}
    assert not bad, "gradients of step 0, relative 2-norm error per tensor: " + json.dumps(dict(sorted(worst.items(), key=lambda kv: -kv[1])[:12]), indent=1)


@pytest.mark.parametrize("weights,learner", [("ipc", "streamed"), ("gloo", "dropin"), ("ipc", "streamed+compact_wire")])
def test_two_learners_two_engines_on_one_gpu_match_one_learner(libprl, cuda_device, tmp_path, weights, learner):
    """BASELINE configs[2]'s topology (2 learner ranks + 2 engines) with every stage on this ONE GPU: six processes, the real HIP preprocessor
    and the real fused-head loss on both learner ranks, gradients all-reduced over gloo (RCCL refuses two ranks on one device), weights
    handed to BOTH engines - as HIP IPC handles each engine maps, or through the weight-update group of 3 over gloo - after every
    optimizer step.  Step 0 equals the single-learner pipeline on the same rollouts: the ranks' partial losses add up to its loss, the
    aggregated statistics agree, and the SGD update is its update / 2 (DDP averages the ranks' gradients like the reference's engines do;
    the loss normaliser is the global samples_per_step on every rank, finetune_loop.py:644-646).  Dense rollouts: one sequence per
    micro-batch on either side, so the two runs differ in WHO trains a sequence, not in how it is packed."""
    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    learner, _, wire = learner.partition("+")
    wire = "compact" if wire else "full"  # compact: each rank's loader expands its own partition's records on the shared GPU
    bs, seq, lr, steps = 16, 96, 0.05, 3
    runs = {}
    for tag, n, m in (("two", 2, 2), ("one", 1, 1)):
        exp, cap = tmp_path / tag / "exp", tmp_path / tag / "cap"
        spec = PipelineSpec(exp_path=str(exp), model="tiny", global_batch=bs, seq_length=seq, attempts=4, steps=steps, optimizer="sgd", lr=lr, param_dtype="fp32",
                            capture_step0=str(cap), n_problems=5, concurrent_groups=2, stage_timeout_s=600.0, learner=learner, dense=True, wire=wire,
                            n_learners=n, n_engines=m, weight_transport=weights if n > 1 else "ipc", share_device=True, extra={"bucket_bytes": 1 << 16})
        res = run_pipeline(spec)
        assert "error" not in res, json.dumps(res.get("error"), indent=1)[:6000]
        runs[tag] = (res, cap)
    res, cap2 = runs["two"]
    st, s = res["stages"], res["summary"]
    assert s["topology"]["learners"] == 2 and s["topology"]["engines"] == 2 and s["topology"]["grad_backend"] == "gloo"
    assert s["topology"]["weight_transport"] == {"ipc": "hip_ipc_colocated", "gloo": "gloo_host_staged"}[weights]
    assert set(s["topology"]["devices"].values()) == {"cuda:0"}
    l0, l1 = st["learner0"], st["learner1"]
    assert l0["completed_steps"] == l1["completed_steps"] == steps and l0["local_samples"] == l1["local_samples"] == steps * bs // 2
    assert l0["micro_batches"] == l1["micro_batches"]
    for e in ("engine0", "engine1"):
        assert st[e]["updates"] == steps + 1 and st[e]["last_version"] == steps * bs
        if weights == "gloo":
            assert st[e]["weight_group"]["size"] == 3 and st[e]["weight_group"]["bytes_received"] == l0["weight_group"]["bytes_sent"] > 0
    assert s["engine_weights_equal_trainer_at_last_version"] is True and st["engine0"]["param_probe"] == st["engine1"]["param_probe"]
    assert st["preprocessor"]["published_samples"] >= steps * bs
    a0, a1 = torch.load(cap2 / "rank0" / "params_after.pt"), torch.load(cap2 / "rank1" / "params_after.pt")
    assert all(torch.equal(a0[n], a1[n]) for n in a0), "DDP kept the replicas identical"
    # ---- against the single-learner pipeline
    _, cap1 = runs["one"]
    b1, b2 = torch.load(cap1 / "params_before.pt"), torch.load(cap2 / "rank0" / "params_before.pt")
    assert all(torch.equal(b1[n], b2[n]) for n in b1)
    ids = lambda captured: sorted(tuple(b["input_ids"].flatten().tolist()) for b in captured if not b["sentinel"])  # noqa: E731
    assert ids(torch.load(cap1 / "step0_batches.pt")) == ids(torch.load(cap2 / "rank0" / "step0_batches.pt") + torch.load(cap2 / "rank1" / "step0_batches.pt"))
    m1, m2 = json.loads((cap1 / "step0_metrics.json").read_text()), json.loads((cap2 / "rank0" / "step0_metrics.json").read_text())
    assert m2["rl/loss"] == pytest.approx(m1["rl/loss"], rel=1e-4, abs=1e-7)
    for k, w in m1.items():
        if k.startswith("rl/"):
            assert m2[k] == pytest.approx(w, rel=1e-4, abs=1e-6), k
    after1 = torch.load(cap1 / "params_after.pt")
    g1, g2 = torch.load(cap1 / "grads_step0.pt"), torch.load(cap2 / "rank0" / "grads_step0.pt")
    worst = {}
    for n in g1:
        want, got = 0.5 * g1[n].double(), g2[n].double()
        worst[n] = float((got - want).norm() / max(float(want.norm()), 1e-30))
        d1, d2 = (after1[n] - b1[n]).double(), (a0[n] - b2[n]).double()
        # (an update is ~1e-5 of a weight: each side carries half an ulp of the fp32 parameter it was subtracted from)
        ulp = torch.finfo(torch.float32).eps * torch.maximum(b1[n].abs(), after1[n].abs()).double()
        assert bool(((d2 - 0.5 * d1).abs() <= 1.5 * ulp + 1e-3 * float(d1.abs().max())).all()), f"{n}: update of the 2-rank run = update of the 1-rank run / 2"
    bad = {n: e for n, e in worst.items() if e > 1e-4}
    assert not bad, "averaged gradients of step 0 vs half the single-learner gradients, relative 2-norm error: " + json.dumps(bad, indent=1)


def test_pipeline_with_reference_policy_compact_wire_equals_full_wire(libprl, cuda_device, tmp_path):
    """BASELINE configs[4]'s loss configuration (KL-to-reference on, kl_coef 0.001) through the pipeline on BOTH wires: the preprocessor holds the
    frozen reference policy and writes `ref_logprobs`; on the compact wire that column is the only per-token data that leaves its GPU (4 B/token,
    `ref_column`).  Step 0's micro-batches are identical on the two wires - every column bit for bit, the reference column different from the
    rollout log-probs - and so are the loss and the KL statistics."""
    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    bs, seq, steps = 16, 96, 2
    got = {}
    for wire in ("full", "compact"):
        exp, cap = tmp_path / wire / "exp", tmp_path / wire / "cap"
        spec = PipelineSpec(exp_path=str(exp), model="tiny", global_batch=bs, seq_length=seq, attempts=4, steps=steps, optimizer="sgd", lr=0.05, param_dtype="fp32",
                            capture_step0=str(cap), n_problems=5, concurrent_groups=2, stage_timeout_s=600.0, learner="streamed", wire=wire, kl_coef=0.001,
                            ref_seed=4242)  # (a reference policy that differs from the initial policy: the KL term is non-zero from step 0 on)
        res = run_pipeline(spec)
        assert "error" not in res, json.dumps(res.get("error"), indent=1)[:6000]
        assert res["summary"]["optimizer_steps"] == steps and res["summary"]["engine_weights_equal_trainer_at_last_version"] is True
        assert res["stages"]["preprocessor"]["host_phase_s"].get("ref_logprobs", 0) > 0, "the reference policy ran in the preprocessor"
        got[wire] = (torch.load(cap / "step0_batches.pt"), json.loads((cap / "step0_metrics.json").read_text()))
    (fb, fm), (cb, cm) = got["full"], got["compact"]
    assert len(fb) == len(cb) > 0
    for a, b in zip(fb, cb):
        assert a.keys() == b.keys()
        for k in a:
            if torch.is_tensor(a[k]):
                assert torch.equal(a[k], b[k]), k
            else:
                assert a[k] == b[k], k
        if not a["sentinel"]:
            lab = a["labels"] != -100
            assert bool((a["ref_logprobs"][lab] != a["old_logprobs"][lab]).any()) and bool((a["ref_logprobs"][~lab] == 0).all())
    assert fm["rl/loss"] == cm["rl/loss"] and fm["rl/kl"] == cm["rl/kl"] and abs(fm["rl/kl"]) > 0


def test_pipeline_with_a_tensor_parallel_engine_on_one_gpu(libprl, cuda_device, tmp_path):
    """BASELINE configs[4]'s engine layout in the pipeline, on this one GPU: two learner ranks, one TP = 2 engine (two inference workers holding
    vLLM-style stacked slices: qkv_proj, gate_up_proj, row-cut o_proj / down_proj, vocabulary-cut embeddings), KL-to-reference on.  After every
    optimizer step each TP rank receives ITS slices through the group of its rank (gloo here; `weight_transport="rccl"` on real GPUs) -
    about half of the parameter bytes - and ends with exactly the trainer's slices."""
    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    bs, steps = 16, 2
    spec = PipelineSpec(exp_path=str(tmp_path / "exp"), model="tiny", global_batch=bs, seq_length=96, attempts=4, steps=steps, n_problems=5, concurrent_groups=2,
                        stage_timeout_s=600.0, learner="streamed", n_learners=2, n_engines=1, engine_tp=2, weight_transport="gloo", share_device=True,
                        kl_coef=0.001, extra={"bucket_bytes": 1 << 16})
    res = run_pipeline(spec)
    assert "error" not in res, json.dumps(res.get("error"), indent=1)[:6000]
    st, s = res["stages"], res["summary"]
    assert s["topology"]["engine_tp"] == 2 and s["optimizer_steps"] == steps and s["engine_weights_equal_trainer_at_last_version"] is True
    e, l0 = st["engine"], st["learner0"]
    assert e["updates"] == steps + 1 and e["weight_group"]["ranks"] == [1, 2] and e["weight_group"]["size"] == 2
    sent = l0["weight_group"]["bytes_sent_per_tp_rank"]
    total = l0["weight_group"]["param_bytes"] * (steps + 1)
    assert e["weight_group"]["bytes_received_per_tp_rank"] == sent and all(0.45 * total < b < 0.65 * total for b in sent), (sent, total)
    want = l0["param_probes_per_tp_rank"][str(steps * bs)]
    assert e["param_probe"] == want and want[0] != want[1] and len(want[0]) == 3  # final norm (replicated), q_proj rows, embedding rows
    assert st["preprocessor"]["host_phase_s"].get("ref_logprobs", 0) > 0
