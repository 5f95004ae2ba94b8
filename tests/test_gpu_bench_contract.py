"""bench.py prints ONE JSON line that satisfies the driver's contract (run on the smallest
workload so the test takes seconds)."""

import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(extra, tmp_path, timeout=900):
    detail = tmp_path / "bench_detail.json"
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--detail-out", str(detail), *extra],
                         capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert out.stdout.rstrip().splitlines()[-1] == lines[0]  # the line is the LAST thing on stdout (the driver parses the tail)
    return lines[0], json.loads(detail.read_text())


def test_bench_json_contract(libprl, cuda_device, tmp_path):
    """The default invocation: one line under 4 KB with the driver's keys, `roofline` and `cpu_baseline` scalar-valued, the rest in the detail file."""
    text, full = _run([], tmp_path)
    assert len(text.encode()) < 4096
    d = json.loads(text)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert type(d[key]) is typ, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] in ("strong", "weak")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] / 1e3)) <= 1e-6 * d["value"]
    # `value` is timed on the reference's behaviour (every row read, rl/__init__.py:213); the opt-out is a separate, labelled number
    assert d["config"]["skip_unlabelled"] is False and d["value_skip_unlabelled"] > 0 and d["skip_unlabelled_steps"] == 1
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r and "traffic_source" in r
    # the line alone reproduces `achieved`: algorithmic bytes per launch / average launch duration (HIP events)
    assert r["launches"] == d["steps"] * d["config"]["global_batch"] and r["avg_us"] >= r["min_us"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9) <= 1e-9 * r["achieved"]
    V, T = d["config"]["vocab"], d["config"]["seq_len"]
    assert r["algorithmic_bytes_per_launch"] == T * (2 * V * 4 + 56)  # every row: logits read once + d logits written once
    c = d["cpu_baseline"]
    assert type(c["cores"]) is int and c["cores"] >= 1 and c["kind"] == "port" and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    w = d["weight_sync"]  # N = 1: colocated hand-off over HIP IPC
    assert "error" not in w, w
    assert w["metric"] == "trainer_to_actor_weight_sync_ms" and w["median_ms"] > 0 and w["gbytes"] > 0.9 and w["transport"] == "hip_ipc_colocated"
    assert "excludes the transformer" in d["metric"]
    assert {"fused_logits_loss", "grpo_loss_step", "preprocess_K5_K6", "pack_collate_kernel", "group_advantages_K5"} <= set(d["hbm_frac"])
    for key in ("kernels", "roofline_mfma", "pipeline", "e2e", "preprocess_loop", "ref_logprob", "transport"):
        assert key not in d
    # ---- the detail file: the same run, everything that was measured
    assert full["value"] == d["value"] and full["roofline"]["avg_us"] == r["avg_us"]
    k = full["kernels"]
    assert k["pack_collate_kernel"]["avg_us"] <= k["preprocess_K5_K6"]["avg_us"]  # the kernel alone vs kernel + host planning
    o = full["value_skip_unlabelled"]
    assert o["kernel"]["algorithmic_bytes"] < r["algorithmic_bytes_per_launch"] and o["value"] == d["value_skip_unlabelled"]
    # BASELINE.md §2, leg by leg: the port on this box next to the reference's own function (a committed constant)
    legs = full["cpu_baseline"]["legs"]
    assert {"preprocess", "collate_packed", "wire", "loss_v8", "logprob_fwd", "logprob_fwd_bwd_closed_form"} <= set(legs)
    assert all(leg["us_per_token"] > 0 for leg in legs.values())
    ref = full["cpu_baseline"]["reference"]
    assert ref is None or (ref["source"].startswith("profiles/") and all(leg["reference_us_per_token"] > 0 for leg in legs.values()))
    assert "pipeline" not in full and "roofline_mfma" not in full  # side measurements are --detail only


def test_bench_detail_side_measurements(libprl, cuda_device, tmp_path):
    """`--detail`: the side measurements run and land in the detail file; the printed line keeps its shape."""
    text, d = _run(["--detail"], tmp_path, timeout=1500)
    assert len(text.encode()) < 4096 and "pipeline" not in json.loads(text)
    m = d["roofline_mfma"]  # the MFMA-bound fused head, a second roofline object
    assert "error" not in m, m
    assert m["bound"] == "mfma" and m["unit"] == "TFLOP/s" and m["peak"] == 2500.0 and abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-12
    assert m["config"]["logits_materialised_bytes"] == 0 and m["backward"]["ms"] > 0
    t = d["transport"]  # shm log vs files backend, host side
    assert "error" not in t, t
    assert 0 < t["shm_us_per_token"] < t["files_us_per_token"] and 60 < t["shm_bytes_per_token"] < 80 < t["files_bytes_per_token"]
    assert t["rollout_record"]["PRLROL01_bytes_per_token"] < t["rollout_record"]["jsonl_bytes_per_token"]
    p = d["preprocess_loop"]  # actor records -> PreprocessorLoop -> published micro-batches, at chunk_n_groups = 2
    assert "error" not in p, p
    fast, slow, text_case = (p["cases"][k] for k in ("PRLROL01_to_shm_1_trainer", "PRLROL01_to_shm_1_trainer_one_copy_per_array", "JSONL_to_files_1_trainer"))
    assert fast["tokens_per_s"] > text_case["tokens_per_s"] > 0 and fast["published_samples"] == slow["published_samples"]
    assert fast["transfers_per_chunk"]["h2d"] <= 2.5 and fast["transfers_per_chunk"]["d2h"] <= 1.5  # one upload per chunk + one K6 plan, one download
    assert {"K5", "K6"} <= set(fast["kernel_us_per_chunk"]) and 0 < fast["host_planning_frac"] < 1
    cmp = p["cases"]["PRLROL01_to_shm_1_trainer_compact_wire"]  # the micro-batch before expansion on the wire; K6 in the learner's loader
    assert cmp["published_samples"] == fast["published_samples"] and "K6" not in cmp["kernel_us_per_chunk"]
    assert cmp["consumer"]["tokens"] == fast["consumer"]["tokens"] > 0
    assert cmp["consumer"]["log_bytes_per_token"] < 20 < 60 < fast["consumer"]["log_bytes_per_token"] < 80
    q = d["ref_logprob"]  # reference-policy head: stock GEMM + [T, V] logits + K1 vs the MFMA head
    assert "error" not in q, q
    for name, h in q["heads"].items():
        # the old bf16 path rounds the LOGITS to bf16 (2^-9 of |logit|): its log-probs are the less exact ones
        assert h["fused_ms"] > 0 and h["old_ms"] > 0 and h["max_abs_difference"] < (1e-3 if name == "fp32_head" else 0.1) and h["hbm_bytes_saved"] > 0
    pl = d["pipeline"]  # the four stages as processes (here around the two-layer model)
    assert "error" not in pl, pl
    assert pl["samples_per_s"] > 0 and pl["optimizer_steps"] >= 2 and pl["engine_weights_equal_trainer_at_last_version"] is True
    assert set(pl["busy_frac"]) == {"actor", "preprocessor", "learner", "engine"} and pl["weight_sync_under_load_ms"]["updates"] == pl["optimizer_steps"]
    assert pl["overlap"]["pipelined_s_per_step"] > 0 and sum(pl["lag_optimizer_steps_histogram"].values()) > 0


@pytest.mark.parametrize("world", [2, 8])
def test_bare_multi_gpu_invocation_becomes_n_ranks(libprl, cuda_device, world):
    """`python bench.py --gpus N` with no launcher re-executes itself as N ranks (here all on this GPU over gloo, the dry-run mode)
    and the line's `n_gpus` comes from the process group, not from argv.  N = 8 is the driver's scaling run: the batch splits into
    whole groups per rank, every rank shards / packs / runs its share, the statistics all-gather closes the step."""
    import os

    env = {**os.environ, "PRL_BENCH_SHARE_DEVICE": "1"}
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--backend", "gloo", "--workload", "tiny", "--steps", "1", "--warmup", "0",
                          "--no-weight-sync", "--detail-out", "gpurun_out/bench_detail_2ranks.json"], capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert len(lines[0]) < 4096
    td = json.loads((ROOT / "gpurun_out" / "bench_detail_2ranks.json").read_text())["config_detail"]["torch_distributed"]
    assert d["n_gpus"] == world and td["world_size"] == world and td["backend"] == "gloo" and td["share_device_dry_run"] is True and td["distinct_devices"] == 1
    assert d["config"]["parallelism"] == f"dp{world}" and d["config"]["backend"] == "gloo" and d["config"]["distinct_devices"] == 1 and d["cpu_baseline"] is None
