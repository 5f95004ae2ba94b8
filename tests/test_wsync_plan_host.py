"""The scatter + all-gather transfer plan of `prl_wsync_bcast_bucket_sag` (csrc/prl_wsync_plan.h - the header the RCCL
code executes) compiled for the host and run for every rank of a simulated group: slices tile the bucket on 256-byte
boundaries, every send has its receive (same peer, offset, length, same order between the pair - what a
ncclGroupStart / ncclGroupEnd exchange needs), rank 0 only sends, and executing the table leaves every receiver with
the sender's bytes.  Replaces the reference's one `broadcast` per parameter (finetune_loop.py:230-238,
vllm1.py:110-127); no box with two GPUs has run the RCCL side yet, so the arithmetic is pinned here."""

import ctypes
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "harness" / "wsync_plan_host.cpp"
OUT = ROOT / "tests" / "harness" / "libwsync_plan_host.so"


@pytest.fixture(scope="module")
def plan():
    hdr = ROOT / "pipelinerl_amd" / "csrc" / "prl_wsync_plan.h"
    if not OUT.exists() or OUT.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(OUT), str(SRC)])
    lib = ctypes.CDLL(str(OUT))
    lib.prl_wsync_plan_check.restype = ctypes.c_int
    lib.prl_wsync_plan_check.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
    lib.prl_wsync_plan_slice_lo.restype = ctypes.c_uint64
    lib.prl_wsync_plan_slice_lo.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
    return lib


@pytest.mark.parametrize("receivers", range(1, 8))
def test_plan_moves_every_byte_to_every_receiver(plan, receivers):
    world = receivers + 1
    sizes = {1, 255, 256, 257, 4096, 65536 + 3, 256 * receivers - 1, 256 * receivers, 256 * receivers + 1,
             256 * receivers * 3 + 129, (1 << 20) + 3}
    for n in sorted(sizes):
        assert plan.prl_wsync_plan_check(world, n, 1) == 0, (receivers, n)


@pytest.mark.parametrize("receivers", range(1, 8))
def test_plan_of_a_gigabyte_bucket_tiles_without_executing(plan, receivers):
    n = (1 << 30) + 3
    assert plan.prl_wsync_plan_check(receivers + 1, n, 0) == 0
    los = [plan.prl_wsync_plan_slice_lo(n, receivers, i) for i in range(receivers + 1)]
    assert los[0] == 0 and los[-1] == n and all(a <= b for a, b in zip(los, los[1:]))
    assert all(lo % 256 == 0 for lo in los[:-1])
    # balanced: no slice is more than 256 bytes above the mean
    assert max(b - a for a, b in zip(los, los[1:])) <= -(-n // receivers) + 255


def test_small_buckets_leave_trailing_slices_empty(plan):
    # 300 bytes on 7 receivers: 256 + 44, five empty slices - and still a complete broadcast
    los = [plan.prl_wsync_plan_slice_lo(300, 7, i) for i in range(8)]
    assert los == [0, 256, 300, 300, 300, 300, 300, 300]
    assert plan.prl_wsync_plan_check(8, 300, 1) == 0
    assert plan.prl_wsync_plan_check(1, 123, 1) == 0  # a group of one: nothing to do
    assert plan.prl_wsync_plan_check(4, 0, 1) == 0
