"""The ONE line `bench.py` prints (`bench.split_line`): small enough for the driver's stdout tail, scalars where the contract has
scalars, strict JSON.  Round 5's line had grown to 30 KB with `cpu_baseline.cores` a dict and the driver recorded `parsed: null`."""

import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SCALARS = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float, "higher_is_better": bool,
           "scaling": str, "dtype": str, "data": str}


def _full(long: int = 1):
    """What main() hands to split_line, with every free-text field blown up `long` times."""
    pad = "x" * (400 * long)
    kernels = {n: {"launches": 20, "avg_us": 100.0, "min_us": 90.0, "algorithmic_bytes": 1e9, "GBps": 5000.0, "hbm_frac": 0.625, "note": pad}
               for n in ("group_advantages_K5_kernels", "group_advantages_K5", "pack_collate_kernel", "preprocess_K5_K6", "fused_logits_loss", "grpo_loss_step")}
    return {
        "metric": "learner samples/sec, 7B GRPO bs=4096 (post-model hot path ...)", "value": 545.1, "unit": "samples/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 7514.2, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "7b_grpo_bs4096_seq8192", "global_batch": 4096, "seq_len": 8192, "vocab": 152064, "tokens_per_step": 33554432, "parallelism": "dp1",
                   "logits_mode": "fused", "policy_loss": "ppo", "kl_coef": 0.0, "skip_unlabelled": False, "labelled_token_fraction": 0.9647,
                   "grad_allreduce_bytes_per_step": 0, "backend": None, "distinct_devices": 1},
        "config_detail": {"note": pad},
        "roofline": {"bound": "hbm", "kernel": "fused_logits_loss", "achieved": 5500.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.6875, "traffic": 1.08e10,
                     "traffic_source": "committed: " + pad, "avg_us": 1812.0, "min_us": 1760.0, "launches": 81920, "algorithmic_bytes_per_launch": 9965725696.0,
                     "timing": pad},
        "value_skip_unlabelled": {"value": 557.0, "steps": 2, "ms_per_step": 7350.0, "what": pad, "kernel": {"avg_us": 1790.0}},
        "kernels": kernels,
        "cpu_baseline": {"value": 0.2, "unit": "samples/s", "cores": 16, "kind": "port", "measured_in_this_run": True, "sample": "oracle ... " + pad,
                         "legs": {k: {"us_per_token": 1.0, "what": pad} for k in ("preprocess", "collate_packed", "wire", "loss_v8", "logprob_fwd")},
                         "host": {"nproc": 256, "cgroup_cpu_quota": 16},
                         "reference": {"samples_per_s_extrapolated": 0.1088, "threads": 8, "source": "profiles/r03_reference_cpu_legs.json", "legs": {"a": pad}}},
        "weight_sync": {"metric": "trainer_to_actor_weight_sync_ms", "transport": "hip_ipc_colocated", "median_ms": 8.7, "min_ms": 8.6, "gbytes": 15.231, "tensors": 339,
                        "effective_GBps": 1752.0, "transport_note": pad, "layout": pad},
        "skipped": {"live_pmc": "not run: " + pad},
        "loss": 1.0887, "wall_s": 240.0,
        "roofline_mfma": {"note": pad}, "pipeline": {"what": pad}, "e2e": {"note": pad}, "preprocess_loop": {"what": pad}, "ref_logprob": {"what": pad},
        "transport": {"what": pad},
    }


def _strict_loads(text: str):
    def no_constants(name):
        raise ValueError(f"non-standard JSON constant {name}")

    return json.loads(text, parse_constant=no_constants)


@pytest.mark.parametrize("long", [1, 40])
def test_line_is_small_scalar_and_strict_json(long):
    import bench

    line, detail = bench.split_line(_full(long), "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text.encode()) < bench.LINE_LIMIT_BYTES == 4096 and "\n" not in text
    d = _strict_loads(text)
    for key, typ in SCALARS.items():
        assert type(d[key]) is typ, (key, type(d[key]))  # no nested object where rounds 1-4 had a scalar
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert all(not isinstance(v, (dict, list)) for v in d["config"].values())
    r = d["roofline"]
    assert set(r) == {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_us", "min_us", "launches", "algorithmic_bytes_per_launch"}
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    c = d["cpu_baseline"]
    assert type(c["cores"]) is int and c["kind"] in ("port", "reference") and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert all(not isinstance(v, (dict, list)) for v in c.values())
    assert c["reference_value"] == 0.1088 and c["reference_cores"] == 8  # the committed constant rides along as scalars
    w = d["weight_sync"]
    assert w["transport"] == "hip_ipc_colocated" and w["median_ms"] == 8.7 and w["gbytes"] == 15.231 and "transport_note" not in w
    assert type(d["value_skip_unlabelled"]) is float and d["skip_unlabelled_steps"] == 2
    assert d["detail"] == "gpurun_out/bench_detail.json"
    # none of the side measurements is on the line; all of them are in the detail object
    for key in ("kernels", "roofline_mfma", "pipeline", "e2e", "preprocess_loop", "ref_logprob", "transport", "config_detail"):
        assert key not in d and key in detail
    assert detail["cpu_baseline"]["legs"] and set(d["hbm_frac"]) == set(detail["kernels"])


def test_line_without_optional_legs():
    """N > 1 ranks have no CPU baseline; a run may skip the weight-sync leg and the opt-out."""
    import bench

    full = _full()
    full.update(cpu_baseline=None, weight_sync=None, value_skip_unlabelled=None, skipped={})
    line, _ = bench.split_line(full, "x.json")
    assert line["cpu_baseline"] is None and line["weight_sync"] is None and line["value_skip_unlabelled"] is None and "skipped" not in line
    _strict_loads(json.dumps(line))


def test_multi_gpu_line_carries_the_exchange_step_and_the_rccl_broadcast():
    import bench

    full = _full()
    full["n_gpus"] = 8
    full["kernels"]["grad_allreduce"] = {"launches": 20, "avg_us": 60000.0, "min_us": 59000.0, "bytes": 15231233024, "algbw_GBps": 253.0, "busbw_GBps": 443.0}
    full["weight_sync"] = {"metric": "trainer_to_actor_weight_sync_ms", "transport": "rccl_xgmi", "median_ms": 90.0, "gbytes": 15.231, "tensors": 339, "n_receivers": 7,
                           "broadcast_ms": 120.0, "scatter_allgather_ms": 80.0, "full_update_verified": True, "stage": "done", "rccl_comm_size": 8}
    line, _ = bench.split_line(full, "x.json")
    assert line["grad_allreduce"] == {"avg_ms": 60.0, "bytes": 15231233024, "busbw_GBps": 443.0}
    assert line["weight_sync"]["transport"] == "rccl_xgmi" and line["weight_sync"]["n_receivers"] == 7 and line["weight_sync"]["full_update_verified"] is True
    assert len(json.dumps(line)) < 4096


def test_an_oversized_line_is_refused_not_printed():
    import bench

    full = _full()
    full["config"]["workload"] = "w" * 5000
    with pytest.raises(AssertionError, match="bytes"):
        bench.split_line(full, "x.json")


def test_gradient_buckets_of_the_baseline_models():
    """N > 1: the data-parallel exchange step all-reduces the model's bf16 gradients in 1 GiB buckets - sized here, for every
    BASELINE workload, without a GPU."""
    import bench

    for name, (_, _, _, nbytes) in bench.WORKLOAD_MODEL.items():
        sizes = bench.grad_bucket_sizes(nbytes)
        assert sum(sizes) == nbytes and all(0 < n <= 1 << 30 and n % 2 == 0 for n in sizes) and all(n == 1 << 30 for n in sizes[:-1]), name
    assert len(bench.grad_bucket_sizes(bench.WORKLOAD_MODEL["7b_grpo_bs4096_seq8192"][3])) == 15
    assert len(bench.grad_bucket_sizes(bench.WORKLOAD_MODEL["32b_grpo_kl_bs4096_seq8192"][3])) == 62
    assert bench.grad_bucket_sizes(0) == []
    with pytest.raises(ValueError):
        bench.grad_bucket_sizes(3)
    # every BASELINE batch splits into whole groups per rank at 1, 2, 4 and 8 ranks
    for name, (bs, _, _, attempts) in bench.WORKLOADS.items():
        for world in (1, 2, 4, 8):
            assert bs % (attempts * world) == 0, (name, world)
