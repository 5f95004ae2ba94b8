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


def test_ipc_transport_cuts_tensors_too_large_for_one_exportable_allocation(monkeypatch):
    """`hipIpcOpenMemHandle` does not return for allocations of 2 GiB and more on this stack (profiles/r05r_ipc_open_by_allocation_size.txt):
    the IPC transport turns such a tensor into row ranges on both sides (same arithmetic from `parameters_info`), no bucket reaches the
    limit, and a receiver without a registered destination gets the tensor back in one piece."""
    import torch

    from pipelinerl_amd.weight_sync import (IPC_MAX_ALLOCATION, ColocatedReceiver, ParamSpec, bucket_nbytes, plan_buckets, rows_piece, split_for_ipc)

    # the real case: a 7B fp32 output head
    specs = [ParamSpec("model.norm.weight", (3584,), torch.bfloat16), ParamSpec("lm_head.weight", (152064, 3584), torch.float32)]
    cut = split_for_ipc(specs)
    assert cut[0] == specs[0] and [rows_piece(s.name) for s in cut[1:]] == [("lm_head.weight", 0, 74898), ("lm_head.weight", 74898, 149796), ("lm_head.weight", 149796, 152064)]
    assert sum(s.shape[0] for s in cut[1:]) == 152064 and all(s.shape[1:] == (3584,) and s.dtype == torch.float32 for s in cut[1:])
    assert all(bucket_nbytes(b) < IPC_MAX_ALLOCATION for b in plan_buckets(cut))
    assert split_for_ipc(specs[:1]) == specs[:1] and rows_piece("a#rowsx:3") is None and rows_piece("plain.name") is None
    assert len(split_for_ipc([ParamSpec("flat", (1 << 30,), torch.float32)])) == 4  # a 4 GiB vector: element ranges of 1 GiB
    with pytest.raises(ValueError, match="no leading dimension"):
        split_for_ipc([ParamSpec("one_row", (1, 1 << 30), torch.float32)])

    # small limits: the receiver's side of the arithmetic with hand-filled buckets (no device needed on the load_weights path)
    torch.manual_seed(0)
    tensors = {"a": torch.randn(5, 8), "big": torch.randn(100, 16), "z": torch.randn(3)}
    info = [ParamSpec(n, tuple(t.shape), t.dtype) for n, t in tensors.items()]
    limit, piece = 2048, 1024
    cut = split_for_ipc(info, piece, limit)
    assert [s.name for s in cut][:3] == ["a", "big#rows0:16", "big#rows16:32"] and len(cut) == 2 + 7
    plan = plan_buckets(cut, piece)
    bufs = []
    for bucket in plan:
        buf = torch.zeros(bucket_nbytes(bucket), dtype=torch.uint8)
        for sp, off in bucket:
            p = rows_piece(sp.name)
            src = tensors[p[0]][p[1]:p[2]] if p else tensors[sp.name]
            buf[off:off + sp.nbytes] = src.contiguous().view(torch.uint8).reshape(-1)
        bufs.append(buf)
    rx = ColocatedReceiver(torch.device("cpu"), piece, limit)
    rx._mapped = {f"{k:02x}": type("M", (), {"tensor": (lambda self, b=b: b), "close": lambda self: None})() for k, b in enumerate(bufs)}
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)  # the CPU "device" has nothing to wait for
    got = {}
    n = rx.receive(info, [f"{k:02x}" for k in range(len(bufs))], [b.numel() for b in bufs], lambda views: got.update({k: v.clone() for k, v in views}))
    assert n == 3 and set(got) == set(tensors) and all(torch.equal(got[k], tensors[k]) for k in tensors)
    with pytest.raises(ValueError, match="IPC buckets announced"):
        rx.receive(info, ["00"], [1], lambda v: None)


def test_receiver_refuses_an_update_cut_with_another_allocation_cap():
    """Sender and receiver derive the row-range piece list from (bucket_bytes, ipc_max_allocation).  A receiver that does not use the
    sender's cap either gets another bucket COUNT or - worse - the same count with other cuts; both are refused before any byte is
    mapped (no GPU needed to get that far), and the cap announced in the request makes the lists agree."""
    import torch

    from pipelinerl_amd.weight_sync import ColocatedReceiver, ParamSpec, bucket_nbytes, plan_buckets, split_for_ipc

    info = [ParamSpec("small", (7, 9), torch.float32), ParamSpec("big", (300, 16), torch.float32), ParamSpec("tail", (40, 8), torch.bfloat16)]
    bucket_bytes, sender_cap = 4096, 8192
    sender_plan = plan_buckets(split_for_ipc(info, bucket_bytes, sender_cap), bucket_bytes)
    handles, sizes = ["00"] * len(sender_plan), [bucket_nbytes(b) for b in sender_plan]
    rx = ColocatedReceiver(torch.device("cpu"), bucket_bytes)  # built like vllm_worker builds it: the default cap (2 GiB - 1 MiB)
    with pytest.raises(ValueError, match="IPC buckets announced|disagree"):
        rx.receive(info, handles, sizes, None)
    # same NUMBER of buckets, other cuts inside them (one large bucket; 124-row vs 89-row pieces): the case that would scatter rows wrongly
    wide = 1 << 20
    plan_a, plan_b = (plan_buckets(split_for_ipc(info, wide, cap), wide) for cap in (8192, 6000))
    assert len(plan_a) == len(plan_b) == 1 and [sp.name for sp, _ in plan_a[0]] != [sp.name for sp, _ in plan_b[0]]
    rx_wide = ColocatedReceiver(torch.device("cpu"), wide)
    with pytest.raises(ValueError, match="disagree"):
        rx_wide.receive(info, ["00"], [bucket_nbytes(plan_a[0])], None, max_allocation=6000)
    # with the sender's cap the plan matches and the receiver goes on to map the first bucket (which needs the library + a device)
    with pytest.raises(Exception) as e:
        rx.receive(info, handles, sizes, None, max_allocation=sender_cap)
    assert "announced" not in str(e.value) and "disagree" not in str(e.value)
