"""BASELINE.json configs[4] - Qwen2.5-32B GRPO, bs 4096 x seq 8192, TP = 2 inference x 2 actors + 4 learner GPUs, KL-to-ref
on - on the host: the 771-tensor / 65.5 GB parameter set the trainer broadcasts (finetune_loop.py:205-292, one
broadcast per parameter in the reference), its 1 GiB bucket plan (last bucket partial), the TP = 2 cut of it
(kv_heads = 8 >= tp: no replication groups) and the scatter + all-gather slice plan at every bucket size the set produces.
CPU-only; the device side is tests/test_gpu_qwen32b.py."""

import ctypes
import math

import pytest
import torch

from pipelinerl_amd.tp_shard import TpShard, plan_tp_shards
from pipelinerl_amd.weight_sync import ParamSpec, bucket_nbytes, plan_buckets, plan_shard_buckets
from pipelinerl_amd.weight_sync_probe import qwen25_shapes

GIB = 1 << 30


def _specs():
    return [ParamSpec(n, tuple(s), torch.bfloat16) for n, s in qwen25_shapes("32b")]


def test_parameter_set_is_the_32b_one():
    shapes = qwen25_shapes("32b")
    assert len(shapes) == 771  # 1 embedding + 64 layers x 12 + final norm + untied lm_head
    d = dict(shapes)
    assert d["model.embed_tokens.weight"] == (152064, 5120) and d["lm_head.weight"] == (152064, 5120)
    assert d["model.layers.63.self_attn.q_proj.weight"] == (5120, 5120)      # 40 heads x 128
    assert d["model.layers.63.self_attn.k_proj.weight"] == (1024, 5120)      # 8 KV heads x 128
    assert d["model.layers.0.mlp.gate_proj.weight"] == (27648, 5120) and d["model.layers.0.mlp.down_proj.weight"] == (5120, 27648)
    n_params = sum(math.prod(s) for _, s in shapes)
    assert n_params == 32_763_876_352
    assert 2 * n_params == pytest.approx(65.5e9, rel=1e-3)  # "cfg5: ~65.5 GB, 771 msgs"
    # the smaller sets did not move
    assert len(qwen25_shapes("7b")) == 339 and len(qwen25_shapes("0p5b")) == 290


def test_gibibyte_bucket_plan_of_the_32b_set():
    specs = _specs()
    buckets = plan_buckets(specs, GIB)
    names = [sp.name for b in buckets for sp, _ in b]
    assert names == [sp.name for sp in specs]  # order-preserving: both sides derive the plan from parameters_info alone
    sizes = [bucket_nbytes(b) for b in buckets]
    # embed_tokens and lm_head (1.557 GB each) are larger than a bucket: a bucket of their own, as large as they are
    assert sizes[0] == 152064 * 5120 * 2 and sizes[-1] >= 152064 * 5120 * 2
    assert all(s <= GIB for s in sizes[1:-1])
    assert all(off % 256 == 0 for b in buckets for _, off in b)
    assert sum(sp.nbytes for sp in specs) <= sum(sizes) < sum(sp.nbytes for sp in specs) + 256 * len(specs)
    assert 64 <= len(buckets) <= 80  # one 975 MB layer per bucket, its norms and biases riding along
    assert sizes[-2] < GIB  # partial buckets exist: the plan must not assume full ones


def test_tp2_cut_of_the_32b_set():
    shapes = qwen25_shapes("32b")
    cuts = plan_tp_shards(shapes, 2, kv_heads=8)
    L = "model.layers.17."
    assert cuts[L + "self_attn.k_proj.weight"] == TpShard(0, 2) and cuts[L + "self_attn.v_proj.bias"] == TpShard(0, 2)
    assert cuts[L + "self_attn.k_proj.weight"].shard_shape((1024, 5120)) == (512, 5120)   # 4 of the 8 KV heads
    assert cuts[L + "self_attn.o_proj.weight"].shard_shape((5120, 5120)) == (5120, 2560)
    assert cuts[L + "mlp.down_proj.weight"].shard_shape((5120, 27648)) == (5120, 13824)
    assert cuts["lm_head.weight"].shard_shape((152064, 5120)) == (76032, 5120)
    assert cuts[L + "input_layernorm.weight"] == TpShard()
    specs = _specs()
    total = sum(sp.nbytes for sp in specs)
    replicated = sum(sp.nbytes for sp in specs if cuts[sp.name].dim is None)
    per_rank = []
    for r in range(2):
        groups, mine = plan_shard_buckets(specs, cuts, r, 2, bucket_bytes=GIB)
        assert len(groups) == len(mine)
        assert all(off % 256 == 0 for b in mine for _, off in b)
        per_rank.append(sum(sp.nbytes for b in mine for sp, _ in b))
    assert per_rank[0] == per_rank[1] == (total - replicated) // 2 + replicated
    assert per_rank[0] < 0.5001 * total  # each TP rank receives half of the 65.5 GB, not all of it as in vllm1.py:110-127


@pytest.mark.parametrize("receivers", [1, 2, 3, 4, 7])
def test_scatter_allgather_slices_at_the_32b_bucket_sizes(receivers):
    """The slice table of `prl_wsync_bcast_bucket_sag` (csrc/prl_wsync_plan.h, compiled for the host) at every distinct
    bucket size of the 65.5 GB update - 1.557 GB vocabulary buckets, ~975 MB layer buckets, the partial tail: slices
    tile the bucket on 256-byte boundaries, every send has its matching receive."""
    from test_wsync_plan_host import OUT, ROOT, SRC  # the harness and its build rule
    import subprocess

    hdr = ROOT / "pipelinerl_amd" / "csrc" / "prl_wsync_plan.h"
    if not OUT.exists() or OUT.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(OUT), str(SRC)])
    lib = ctypes.CDLL(str(OUT))
    lib.prl_wsync_plan_check.restype = ctypes.c_int
    lib.prl_wsync_plan_check.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
    lib.prl_wsync_plan_slice_lo.restype = ctypes.c_uint64
    lib.prl_wsync_plan_slice_lo.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
    sizes = sorted({bucket_nbytes(b) for b in plan_buckets(_specs(), GIB)})
    assert len(sizes) >= 3 and sizes[-1] > GIB
    for n in sizes:
        assert lib.prl_wsync_plan_check(receivers + 1, n, 0) == 0, n
        los = [lib.prl_wsync_plan_slice_lo(n, receivers, i) for i in range(receivers + 1)]
        assert los[0] == 0 and los[-1] == n and all(lo % 256 == 0 for lo in los[:-1])
        assert max(b - a for a, b in zip(los, los[1:])) <= -(-n // receivers) + 255
