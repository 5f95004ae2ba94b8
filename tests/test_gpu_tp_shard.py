"""TP-aware weight update on the device: the trainer's slices (dim 0: contiguous views, dim 1: strided views made
contiguous on the way into the staging buffer) are flattened by the gather kernel, carried bucket by bucket on the
transfer streams and scattered straight into the slice storage of each tensor-parallel rank by the scatter kernel.
One GPU: the communicators are loop-backs (the RCCL version needs three GPUs: tests/test_gpu_multi.py)."""

import json

import pytest
import torch

from test_tp_shard import LoopGroups, qwen_shapes

pytestmark = pytest.mark.gpu


def test_sharded_update_through_the_copy_kernels(libprl, cuda_device):
    from pipelinerl_amd.finetune_loop import ParameterInfo, WeightUpdateRequest
    from pipelinerl_amd.tp_shard import plan_tp_shards, shard_view
    from pipelinerl_amd.vllm_worker import StandaloneShardReceiver
    from pipelinerl_amd.weight_sync import ShardedSender

    tp = 2
    shapes = qwen_shapes(layers=3, hidden=256, inter=704, heads=8, kv_heads=2, head_dim=32, vocab=1000)
    torch.manual_seed(5)
    full = [(n, torch.randn(s, device=cuda_device).to(torch.bfloat16 if i % 3 else torch.float32)) for i, (n, s) in enumerate(shapes)]
    dt = {n: t.dtype for n, t in full}
    cuts = plan_tp_shards(shapes, tp, kv_heads=2)
    loops = LoopGroups(tp)
    for g in loops.groups:
        g.device = cuda_device
    sender = ShardedSender(loops.groups, bucket_bytes=1 << 20)
    for update in range(2):  # the second update reuses the staging buffers
        for g in loops.groups:
            g.sent.clear()
            g.cursor.clear()
        if update:
            full = [(n, t + 1) for n, t in full]
        sender.send(full, cuts)
        torch.cuda.synchronize()
        assert len(loops.groups[0].sent) > 2  # several buckets: the two-stream pipeline was exercised
        req = WeightUpdateRequest(version=update, transport="sharded", bucket_bytes=1 << 20, tp_size=tp,
                                  parameters_info=[ParameterInfo(name=n, shape=list(s), dtype=str(dt[n]), shard_dim=cuts[n].dim, shard_parts=cuts[n].parts)
                                                   for n, s in shapes])
        for t in range(tp):
            w = StandaloneShardReceiver(shapes, lambda n: dt[n], cuda_device, t, tp, kv_heads=2)
            r = loops.groups[t].reader(t)
            r.device = cuda_device
            w.model_update_group, w.tp_rank, w.tp_size = r, t, tp
            w.receive_weight_update(json.dumps(req.model_dump()))
            torch.cuda.synchronize()
            for n, x in full:
                assert torch.equal(w.slices[n], shard_view(x, cuts[n], t, tp)), (update, t, n)
    total = sum(x.numel() * x.element_size() for _, x in full)
    assert all(b < 0.56 * total for b in sender.bytes_sent)


@pytest.mark.parametrize("tp,kv_heads", [(2, 2), (4, 2)], ids=["tp2", "tp4_kv2_replicated"])
def test_sharded_update_into_stacked_engine_storage_on_the_device(libprl, cuda_device, tp, kv_heads):
    """The scatter kernel writes into VIEWS of the engine's stacked `qkv_proj` / `gate_up_proj` storage (row blocks at
    non-zero offsets of a larger allocation); every byte is compared with what vLLM's `load_weights` would build from
    the full tensors (`test_tp_shard._vllm_style_rank_storage`, head arithmetic only)."""
    from test_tp_shard import _vllm_style_rank_storage

    from pipelinerl_amd.finetune_loop import ParameterInfo, WeightUpdateRequest
    from pipelinerl_amd.tp_shard import plan_tp_shards
    from pipelinerl_amd.vllm_worker import StackedShardReceiver
    from pipelinerl_amd.weight_sync import ShardedSender

    heads, head_dim = 8, 32
    shapes = qwen_shapes(layers=3, hidden=256, inter=704, heads=heads, kv_heads=kv_heads, head_dim=head_dim, vocab=1000)
    torch.manual_seed(11)
    full = [(n, torch.randn(s, device=cuda_device).to(torch.bfloat16)) for n, s in shapes]
    cuts = plan_tp_shards(shapes, tp, kv_heads=kv_heads)
    loops = LoopGroups(tp)
    for g in loops.groups:
        g.device = cuda_device
    ShardedSender(loops.groups, bucket_bytes=1 << 19).send(full, cuts)
    torch.cuda.synchronize()
    req = WeightUpdateRequest(version=0, transport="sharded", bucket_bytes=1 << 19, tp_size=tp,
                              parameters_info=[ParameterInfo(name=n, shape=list(s), dtype="torch.bfloat16", shard_dim=cuts[n].dim,
                                                             shard_parts=cuts[n].parts) for n, s in shapes])
    for t in range(tp):
        w = StackedShardReceiver(shapes, lambda n: torch.bfloat16, cuda_device, t, tp, kv_heads=kv_heads)
        for x in w.storage.values():
            x.fill_(7.0)
        r = loops.groups[t].reader(t)
        r.device = cuda_device
        w.model_update_group, w.tp_rank, w.tp_size = r, t, tp
        w.receive_weight_update(json.dumps(req.model_dump()))
        torch.cuda.synchronize()
        want = _vllm_style_rank_storage(dict(full), t, tp, heads, kv_heads, head_dim)
        assert set(want) == set(w.storage)
        for n, x in want.items():
            assert torch.equal(w.storage[n], x), (t, n)
