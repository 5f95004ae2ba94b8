"""`bench.live_pmc_traffic` (the live `roofline.traffic` figure): what it does with the counter files of the two rocprofv3
passes, and that every failure mode returns None (bench.py then quotes the committed figure, labelled as such)."""

import subprocess
import sys
import types
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

KERNEL = "void (anonymous namespace)::fused_logits_loss_keep_kernel<(anonymous namespace)::F32, 1024, 2, 16, 9, true>(Args)"


def _fake_rocprof(rows_of):
    """A stand-in for subprocess.run: writes `<-d>/x/1_counter_collection.csv` for the counter named after --pmc."""
    def run(cmd, **kw):
        counter = cmd[cmd.index("--pmc") + 1]
        out = Path(cmd[cmd.index("-d") + 1]) / "host" / "1_counter_collection.csv"
        rows = rows_of(counter)
        if rows is not None:
            out.parent.mkdir(parents=True)
            out.write_text("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n" + "".join(f'{d},"{k}",{c},{v}\n' for d, k, c, v in rows))
        assert "--kernel-trace" in cmd and "--sys-trace" not in cmd and kw["cwd"] == "/tmp" and kw["env"]["TMPDIR"] == "/tmp"
        return types.SimpleNamespace(returncode=0, stdout="", stderr="")
    return run


@pytest.fixture
def bench(monkeypatch):
    import bench as b

    monkeypatch.setattr(b.shutil if hasattr(b, "shutil") else __import__("shutil"), "which", lambda name: sys.executable)
    return b


def test_counters_are_summed_per_dispatch_and_averaged_over_launches(bench, monkeypatch):
    def rows(counter):
        base = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 3000.0}[counter]
        # two launches of the kernel, each reported in two partial rows (one per XCD group), + another kernel that must be ignored
        return [(1, KERNEL, counter, base), (1, KERNEL, counter, base), (2, KERNEL, counter, base + 100), (2, KERNEL, counter, base + 100),
                (3, "some_other_kernel", counter, 9e9), (1, KERNEL, "GRBM_GUI_ACTIVE", 5e9)]

    monkeypatch.setattr(subprocess, "run", _fake_rocprof(rows))
    got = bench.live_pmc_traffic()
    assert got["launches"] == 2 and got["fetch_kb"] == 2100.0 and got["write_kb"] == 6100.0
    assert got["hbm_bytes_per_launch"] == (2 * 2100.0 + 6100.0) * 1024  # FETCH_SIZE doubled (gfx950), both in KB


@pytest.mark.parametrize("mode", ["no_file", "kernel_absent", "nonzero_exit", "timeout"])
def test_any_failure_falls_back_to_none(bench, monkeypatch, mode):
    def run(cmd, **kw):
        if mode == "timeout":
            raise subprocess.TimeoutExpired(cmd, 1)
        rows = {"no_file": None, "kernel_absent": [(1, "other", cmd[cmd.index("--pmc") + 1], 1.0)]}.get(mode, [(1, KERNEL, cmd[cmd.index("--pmc") + 1], 1.0)])
        r = _fake_rocprof(lambda c: rows)(cmd, **kw)
        r.returncode = 1 if mode == "nonzero_exit" else 0
        return r

    monkeypatch.setattr(subprocess, "run", run)
    assert bench.live_pmc_traffic() is None


def test_no_profiler_inside_a_profiler(bench, monkeypatch):
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: pytest.fail("rocprofv3 must not be started under rocprofv3"))
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    assert bench.live_pmc_traffic() is None
