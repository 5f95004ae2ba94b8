"""prl_bucket_gather / prl_bucket_scatter (csrc/prl_copy.hip): byte-exact against torch copies.

The reference stages the weight update one tensor at a time (finetune_loop.py:262-282,
vllm1.py:110-127); the kernel moves a whole bucket per launch.  Byte work -> bit-exact."""

import pytest
import torch

from pipelinerl_amd.weight_sync import ParamSpec, bucket_nbytes, gather_into_bucket, plan_buckets, scatter_from_bucket

pytestmark = pytest.mark.gpu


def _params(dev, shapes, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for i, (shape, dt) in enumerate(shapes):
        t = torch.randint(-2**31, 2**31 - 1, (max(1, int(torch.tensor(shape).prod()) * dt.itemsize // 4 + 1),), generator=g, dtype=torch.int32)
        n = int(torch.tensor(shape).prod()) if len(shape) else 1
        out[f"p{i}"] = t.view(torch.uint8)[: n * dt.itemsize].clone().view(dt).view(shape).to(dev)
    return out


SHAPES = [((1000, 64), torch.bfloat16), ((63,), torch.float32), ((129, 7), torch.bfloat16), ((1,), torch.float32), ((0,), torch.float32),
          ((70000,), torch.float16), ((3, 5, 7), torch.float32), ((16384, 16), torch.bfloat16), ((33,), torch.uint8)]


@pytest.mark.parametrize("n_rep", [1, 20])  # 20 x 9 = 180 segments: three launches per bucket
def test_gather_scatter_match_torch(libprl, cuda_device, n_rep):
    dev = cuda_device
    tensors = _params(dev, SHAPES * n_rep, seed=n_rep)
    specs = [ParamSpec(n, tuple(t.shape), t.dtype) for n, t in tensors.items()]
    for bucket in plan_buckets(specs, 1 << 20):
        nb = bucket_nbytes(bucket)
        buf = torch.full((nb,), 0xA5, dtype=torch.uint8, device=dev)
        want = buf.clone()
        for sp, off in bucket:
            want[off : off + sp.nbytes] = tensors[sp.name].reshape(-1).view(torch.uint8)
        gather_into_bucket(buf, bucket, tensors)
        assert torch.equal(buf, want)  # padding between slots untouched
        dst = {sp.name: torch.zeros_like(tensors[sp.name]) for sp, _ in bucket}
        scatter_from_bucket(buf, bucket, dst)
        for sp, _ in bucket:
            assert torch.equal(dst[sp.name].reshape(-1).view(torch.uint8), tensors[sp.name].reshape(-1).view(torch.uint8)), sp.name


def test_unaligned_and_noncontiguous_sources(libprl, cuda_device):
    dev = cuda_device
    base = torch.arange(0, 4096, dtype=torch.int32, device=dev).view(torch.uint8)
    sliced = base[3 : 3 + 1000]                       # data_ptr not 16-byte aligned: byte-lane path
    transposed = torch.arange(0, 35, dtype=torch.float32, device=dev).view(5, 7).t()  # made contiguous by the wrapper
    tensors = {"a": sliced, "b": transposed}
    specs = [ParamSpec("a", (1000,), torch.uint8), ParamSpec("b", (7, 5), torch.float32)]
    (bucket,) = plan_buckets(specs, 1 << 20)
    buf = torch.zeros(bucket_nbytes(bucket), dtype=torch.uint8, device=dev)
    gather_into_bucket(buf, bucket, tensors)
    offs = {sp.name: off for sp, off in bucket}
    assert torch.equal(buf[offs["a"] : offs["a"] + 1000], sliced)
    assert torch.equal(buf[offs["b"] : offs["b"] + 140].view(torch.float32).view(7, 5), transposed)
    with pytest.raises(ValueError, match="contiguous"):
        scatter_from_bucket(buf, bucket, {"a": torch.zeros(1000, dtype=torch.uint8, device=dev), "b": torch.zeros(5, 7, device=dev).t()})
    with pytest.raises(ValueError, match="announces"):
        gather_into_bucket(buf, bucket, {"a": sliced, "b": torch.zeros(7, 5, dtype=torch.float16, device=dev)})


def test_large_segment_round_trip(libprl, cuda_device):
    """One 600 MB segment (9 156 chunks) + a 2-byte one; checksum both ways."""
    dev = cuda_device
    big = torch.randint(0, 255, (600_000_002,), dtype=torch.uint8, device=dev)
    tensors = {"big": big[:600_000_000], "tiny": big[600_000_000:]}
    specs = [ParamSpec("big", (600_000_000,), torch.uint8), ParamSpec("tiny", (2,), torch.uint8)]
    (bucket,) = plan_buckets(specs, 1 << 30)
    buf = torch.zeros(bucket_nbytes(bucket), dtype=torch.uint8, device=dev)
    gather_into_bucket(buf, bucket, tensors)
    dst = {"big": torch.zeros(600_000_000, dtype=torch.uint8, device=dev), "tiny": torch.zeros(2, dtype=torch.uint8, device=dev)}
    scatter_from_bucket(buf, bucket, dst)
    assert torch.equal(dst["big"], tensors["big"]) and torch.equal(dst["tiny"], tensors["tiny"])
