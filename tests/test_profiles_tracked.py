"""`profiles/` must reproduce the numbers the documents quote: every tracked rocprofv3 kernel-stats summary parses row by
row to a name + 7 numeric fields (round 3's width-cut summaries had lost Calls / TotalDurationNs / AverageNs of the
dominant kernel - files in that state are kept as `*.truncated.txt`, never as `.csv`), and the condenser that writes them
(`scripts/kernel_stats_summary.py`) keeps every numeric column whatever the length of the kernel's signature."""

import csv
import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
FILES = sorted((ROOT / "profiles").glob("*kernel_stats*.csv"))
COLUMNS = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]


@pytest.mark.parametrize("path", FILES, ids=[f.name for f in FILES])
def test_tracked_kernel_stats_rows_are_complete(path):
    with open(path, newline="") as fh:
        rows = list(csv.reader(fh))
    assert rows[0] == COLUMNS
    assert len(rows) > 1
    for row in rows[1:]:
        assert len(row) == 8, row[0][:80]
        calls, total, avg = int(float(row[1])), float(row[2]), float(row[3])
        assert calls >= 1 and total > 0 and abs(avg - total / calls) <= 1e-6 * avg + 1e-3, row[0][:80]
        for x in row[4:]:
            float(x)


def test_at_least_one_bench_summary_names_the_dominant_kernel_with_its_duration():
    found = False
    for path in FILES:
        if "bench" not in path.name:
            continue
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if "fused_logits_loss" in row["Name"] and float(row["AverageNs"]) > 1e5 and int(float(row["Calls"])) >= 1000:
                    found = True
    assert found


def test_condenser_keeps_the_numbers_of_a_400_character_signature(tmp_path):
    spec = importlib.util.spec_from_file_location("kss", ROOT / "scripts" / "kernel_stats_summary.py")
    kss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kss)
    long_name = ("void (anonymous namespace)::fused_logits_loss_keep_kernel<(anonymous namespace)::F32, 1024, 2, 16, 9, true, "
                 "(anonymous namespace)::DenseOut<(anonymous namespace)::F32> >((anonymous namespace)::RowGeom, (anonymous namespace)::FusedArgs, "
                 "(anonymous namespace)::F32::scalar const*, float, float, (anonymous namespace)::DenseOut<(anonymous namespace)::F32>)")
    assert len(long_name) > 300
    src = tmp_path / "raw.csv"
    with open(src, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(COLUMNS)
        w.writerow(["small_kernel(int)", 10, 1000, 100.0, 0.01, 90, 110, 5.0])
        w.writerow([long_name, 28672, 51520195033, 1796881.802211, 93.24, 1747019, 2657158, 1791479.099093])
    dst = tmp_path / "out.csv"
    assert kss.condense(str(src), str(dst)) == 2
    with open(dst, newline="") as fh:
        rows = list(csv.DictReader(fh))
    assert rows[0]["Name"] == "fused_logits_loss_keep_kernel<F32, 1024, 2, 16, 9, true, DenseOut<F32> >"  # sorted by total time
    assert (int(rows[0]["Calls"]), float(rows[0]["AverageNs"]), int(rows[0]["MaxNs"])) == (28672, 1796881.802211, 2657158)
    assert rows[1]["Name"] == "small_kernel"


def test_the_final_bench_line_recomputes_from_its_own_fields():
    """The committed bench line of the round's final validation (profiles/r05z_bench_default.json) is consistent with itself and with the
    rocprofv3 summary of the same command next to it: value = samples / step time, roofline = algorithmic bytes / event average / peak,
    the tracer's average for the dominant kernel within a few per cent of the HIP-event one, traffic above the algorithmic bytes."""
    import json

    d = json.loads((ROOT / "profiles" / "r05z_bench_default.json").read_text())
    assert d["metric"].startswith("samples_per_s") or "samples" in d["metric"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"].startswith("synthetic")
    assert d["value"] == pytest.approx(4096 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9, rel=1e-6)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.6 < r["frac"] < 0.8
    assert 1.0 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.25  # what the kernel moves beyond the unavoidable bytes: the re-read tail
    assert r["launches"] == 4096 * d["steps"]
    rows = [l for l in (ROOT / "profiles" / "r05z_bench_kernel_stats.csv").read_text().splitlines() if l.startswith('"fused_logits_loss_keep_kernel')]
    assert len(rows) == 1
    avg_ns = float(rows[0].split('",')[1].split(",")[2])
    assert abs(avg_ns * 1e-3 - r["avg_us"]) / r["avg_us"] < 0.06  # the tracer spaces the dispatches: a few per cent faster
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == {"reference": 8, "port": 16} and c["value"] > 0
    p = d["pipeline"]
    assert p["samples_per_s"] == pytest.approx(512 / p["s_per_step"], rel=1e-6) and p["engine_weights_equal_trainer_at_last_version"] is True
    m = d["roofline_mfma"]
    assert m["bound"] == "mfma" and m["peak"] == 2500.0 and m["frac"] == pytest.approx(m["achieved"] / m["peak"], rel=1e-9)


def test_the_round_6_driver_line_is_small_recomputes_and_agrees_with_rocprofv3():
    """profiles/r06f_bench_default.json is the line `python3 bench.py --gpus 1 --steps 20 --warmup 5` printed on an MI355X: under 4 KB, scalar
    `cpu_baseline.cores`, value = samples / step time ON THE REFERENCE'S BEHAVIOUR (every logits row read: algorithmic bytes 2 V 4 + 56 per row),
    roofline = algorithmic bytes / HIP-event average / peak, the rocprofv3 average of the same kernel within a few per cent, live PMC traffic
    above the algorithmic bytes; the detail file written by the same run carries the per-kernel table and the CPU legs."""
    import json

    text = (ROOT / "profiles" / "r06f_bench_default.json").read_text().strip()
    assert len(text.encode()) < 4096 and "\n" not in text
    d = json.loads(text)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["dtype"] == "f32"
    assert d["value"] == pytest.approx(4096 / (d["ms_per_step"] * 1e-3), rel=1e-6) and d["config"]["skip_unlabelled"] is False
    assert d["value_skip_unlabelled"] > d["value"]  # the opt-out reads 3.5 % fewer rows
    r = d["roofline"]
    V, T = d["config"]["vocab"], d["config"]["seq_len"]
    assert r["algorithmic_bytes_per_launch"] == T * (2 * V * 4 + 56) and r["launches"] == 4096 * d["steps"]
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9, rel=1e-6)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.6 < r["frac"] < 0.8 and r["peak"] == 8000.0
    assert r["traffic_source"].startswith("live") and 1.0 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.25
    rows = [l for l in (ROOT / "profiles" / "r06f_bench_kernel_stats.csv").read_text().splitlines() if l.startswith('"fused_logits_loss_keep_kernel')]
    assert len(rows) == 1
    avg_ns = float(rows[0].split('",')[1].split(",")[2])
    assert abs(avg_ns * 1e-3 - r["avg_us"]) / r["avg_us"] < 0.03
    c = d["cpu_baseline"]
    assert type(c["cores"]) is int and c["kind"] == "port" and c["value"] > c["reference_value"] > 0 and c["reference_cores"] == 8
    w = d["weight_sync"]
    assert w["transport"] == "hip_ipc_colocated" and w["gbytes"] > 15 and 0 < w["median_ms"] < 50
    assert d["wall_s"] < 270 and "skipped" not in d
    full = json.loads((ROOT / "profiles" / "r06f_bench_default_detail.json").read_text())
    assert full["value"] == d["value"] and set(d["hbm_frac"]) == {k for k, v in full["kernels"].items() if "hbm_frac" in v}
    assert full["cpu_baseline"]["legs"]["logprob_fwd_bwd_closed_form"]["us_per_token"] > 0


def test_hip_events_and_rocprofv3_agree_inside_one_process():
    """profiles/r06w_bench_under_rocprof.json is the line `bench.py --steps 20 --warmup 5` printed WHILE rocprofv3 traced it, r06w_bench_kernel_stats.csv
    the tracer's summary of that same process: the dominant kernel's average by HIP events (timed steps) and by the tracer (all 25 steps) agree to
    1 %, and the launch counts are the run's."""
    import json

    d = json.loads((ROOT / "profiles" / "r06w_bench_under_rocprof.json").read_text())
    r = d["roofline"]
    rows = [l for l in (ROOT / "profiles" / "r06w_bench_kernel_stats.csv").read_text().splitlines() if l.startswith('"fused_logits_loss_keep_kernel')]
    assert len(rows) == 1
    calls, _, avg_ns = rows[0].split('",')[1].split(",")[:3]
    assert int(calls) == 4096 * (d["steps"] + d["warmup"]) and r["launches"] == 4096 * d["steps"]
    assert abs(float(avg_ns) * 1e-3 - r["avg_us"]) / r["avg_us"] < 0.01
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9, rel=1e-6)
