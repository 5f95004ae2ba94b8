"""bench.py prints ONE JSON line that satisfies the driver's contract (run on the smallest
workload so the test takes seconds)."""

import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_json_contract(libprl, cuda_device):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] in ("strong", "weak")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] / 1e3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    # the line alone reproduces `achieved`: algorithmic bytes per launch / average launch duration (HIP events)
    assert r["launches"] == d["steps"] * d["config"]["global_batch"] and r["avg_us"] >= r["min_us"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9) <= 1e-9 * r["achieved"]
    c = d["cpu_baseline"]
    cores = c["cores"] if isinstance(c["cores"], int) else min(c["cores"].values())  # kind "reference": {"reference": n, "port": m}, both hosts named
    assert c["kind"] in ("port", "reference") and cores >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert "traffic_source" in r  # the counter traffic is a committed figure, labelled as such
    assert {"fused_logits_loss", "grpo_loss_step", "preprocess_K5_K6", "pack_collate_kernel", "group_advantages_K5"} <= set(d["kernels"])
    k = d["kernels"]
    assert k["pack_collate_kernel"]["avg_us"] <= k["preprocess_K5_K6"]["avg_us"]  # the kernel alone vs kernel + host planning
    m = d["roofline_mfma"]  # the MFMA-bound fused head, a second roofline object
    assert "error" not in m, m
    assert m["bound"] == "mfma" and m["unit"] == "TFLOP/s" and m["peak"] == 2500.0 and abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-12
    assert m["config"]["logits_materialised_bytes"] == 0 and m["backward"]["ms"] > 0
    # BASELINE.md §2, leg by leg: the port on this box next to the reference's own function (a committed constant)
    legs = c["legs"]
    assert {"preprocess", "collate_packed", "wire", "loss_v8", "logprob_fwd", "logprob_fwd_bwd_closed_form"} <= set(legs)
    assert all(leg["us_per_token"] > 0 for leg in legs.values())
    ref = c["reference"]
    assert ref is None or (ref["source"].startswith("profiles/") and all(leg["reference_us_per_token"] > 0 for leg in legs.values()))
    t = d["transport"]  # shm log vs files backend, host side
    assert "error" not in t, t
    assert 0 < t["shm_us_per_token"] < t["files_us_per_token"] and 60 < t["shm_bytes_per_token"] < 80 < t["files_bytes_per_token"]
    assert t["rollout_record"]["PRLROL01_bytes_per_token"] < t["rollout_record"]["jsonl_bytes_per_token"]
    p = d["preprocess_loop"]  # actor records -> PreprocessorLoop -> published micro-batches, at chunk_n_groups = 2
    assert "error" not in p, p
    fast, slow, text = (p["cases"][k] for k in ("PRLROL01_to_shm_1_trainer", "PRLROL01_to_shm_1_trainer_one_copy_per_array", "JSONL_to_files_1_trainer"))
    assert fast["tokens_per_s"] > text["tokens_per_s"] > 0 and fast["published_samples"] == slow["published_samples"]
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
    w = d["weight_sync"]  # N = 1: colocated hand-off over HIP IPC
    assert "error" not in w, w
    assert w["metric"] == "trainer_to_actor_weight_sync_ms" and w["median_ms"] > 0 and w["gbytes"] > 0.9 and w["transport"] == "hip_ipc_colocated"
    assert "EXCLUDES the transformer" in d["metric"] and "value_e2e" in d
    pl = d["pipeline"]  # the four stages as processes (here around the two-layer model)
    assert "error" not in pl, pl
    assert pl["samples_per_s"] > 0 and pl["optimizer_steps"] >= 2 and pl["engine_weights_equal_trainer_at_last_version"] is True
    assert set(pl["busy_frac"]) == {"actor", "preprocessor", "learner", "engine"} and pl["weight_sync_under_load_ms"]["updates"] == pl["optimizer_steps"]
    assert pl["overlap"]["pipelined_s_per_step"] > 0 and sum(pl["lag_optimizer_steps_histogram"].values()) > 0


def test_bare_multi_gpu_invocation_becomes_n_ranks(libprl, cuda_device):
    """`python bench.py --gpus 2` with no launcher re-executes itself as 2 ranks (here both on this GPU over gloo, the dry-run mode)
    and the line's `n_gpus` comes from the process group, not from argv."""
    import os

    env = {**os.environ, "PRL_BENCH_SHARE_DEVICE": "1"}
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "tiny", "--steps", "1", "--warmup", "0",
                          "--no-fused-head", "--no-transport", "--no-weight-sync"], capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    td = d["config"]["torch_distributed"]
    assert d["n_gpus"] == 2 and td["world_size"] == 2 and td["backend"] == "gloo" and td["share_device_dry_run"] is True and td["distinct_devices"] == 1
    assert d["config"]["parallelism"] == "dp2" and d.get("pipeline") is None
