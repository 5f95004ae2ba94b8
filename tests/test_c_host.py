"""`include/prl.h` is a C header and libprl.so a C library: a plain C99 host (examples/c_host_demo.c,
gcc, no Python or PyTorch in the process) builds against them (CPU) and, on a GPU box, computes the
group-baseline advantages / label counts / overflow flags that the oracle computes."""

import json
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _build(out: Path) -> None:
    if shutil.which("gcc") is None or not Path("/opt/rocm/include/hip/hip_runtime_api.h").exists():
        pytest.skip("gcc or the HIP headers are not available")
    lib = ROOT / "pipelinerl_amd" / "lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
           str(ROOT / "examples" / "c_host_demo.c"), "-L", str(lib), "-lprl", "-L", "/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_c99_host_compiles_and_links_against_the_abi(libprl, tmp_path):
    _build(tmp_path / "c_host_demo")


@pytest.mark.gpu
def test_c99_host_matches_the_oracle(libprl, cuda_device, tmp_path):
    from oracle import preprocess as opre

    exe = tmp_path / "c_host_demo"
    _build(exe)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = json.loads(res.stdout.strip().splitlines()[-1])

    seqs = [[7, 8, 9, 10, 2], [7, 8, 11], [7, 8, 12, 13, 14, 15], [7, 8, 16, 2]]
    meta = [dict(finish_reason="stop", finished=True), dict(finish_reason="length", finished=False), dict(finished=False), dict(finished=True)]
    entries = []
    for i, (ids, m) in enumerate(zip(seqs, meta)):
        entries.append({"input_ids": ids, "labels": [-100, -100] + ids[2:], "logprobs": [-0.5] * (len(ids) - 2), "ref_logprobs": [],
                        "reward": [1.0, 0.0, 0.0, 1.0][i], "group_id": "g", "metadata": {"model_version": 0, "rollout_index": i, "step_index": 0}, **m})
    want = opre.preprocess_chunk(entries, 2, True)
    for i, e in enumerate(want):
        assert got["advantage"][i] == pytest.approx(e["advantages"][0], rel=1e-12, abs=1e-15)
        assert got["group_tokens"][i] == pytest.approx(e["group_tokens"][0], rel=1e-12)
        assert got["num_labels"][i] == e["num_labels"][0] and got["overflow"][i] == e["overflow"][0]
    assert got["bad_call"] == -22 and got["bad_call_message"]
