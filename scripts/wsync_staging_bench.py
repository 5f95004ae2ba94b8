"""Send-side cost of the bucketed weight update on ONE GPU (world_size 1: the RCCL collectives are
no-ops, what is timed is flattening the Qwen2.5-7B parameter set - 339 tensors, 15.2 GB bf16 -
into 1 GiB buckets and the per-bucket call overhead).  The wire time needs N > 1 GPUs (bench.py)."""

import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pipelinerl_amd.weight_sync import BucketedReceiver, BucketedSender, WeightSyncGroup, plan_buckets, ParamSpec  # noqa: E402


def qwen25_7b_shapes():
    H, I, V, L, KV = 3584, 18944, 152064, 28, 512
    out = [("model.embed_tokens.weight", (V, H))]
    for i in range(L):
        p = f"model.layers.{i}."
        out += [(p + "self_attn.q_proj.weight", (H, H)), (p + "self_attn.q_proj.bias", (H,)), (p + "self_attn.k_proj.weight", (KV, H)),
                (p + "self_attn.k_proj.bias", (KV,)), (p + "self_attn.v_proj.weight", (KV, H)), (p + "self_attn.v_proj.bias", (KV,)),
                (p + "self_attn.o_proj.weight", (H, H)), (p + "mlp.gate_proj.weight", (I, H)), (p + "mlp.up_proj.weight", (I, H)),
                (p + "mlp.down_proj.weight", (H, I)), (p + "input_layernorm.weight", (H,)), (p + "post_attention_layernorm.weight", (H,))]
    out += [("model.norm.weight", (H,)), ("lm_head.weight", (V, H))]
    return out


def main():
    dev = torch.device("cuda", 0)
    shapes = qwen25_7b_shapes()
    params = [(n, torch.empty(s, dtype=torch.bfloat16, device=dev).normal_()) for n, s in shapes]
    nbytes = sum(p.numel() * 2 for _, p in params)
    plan = plan_buckets([ParamSpec(n, tuple(p.shape), p.dtype) for n, p in params])
    print(f"{len(params)} tensors, {nbytes / 1e9:.2f} GB, {len(plan)} buckets of <= 1 GiB")
    grp = WeightSyncGroup._init(WeightSyncGroup._new_uid(), 0, 1, dev)
    sender = BucketedSender(grp)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        specs = sender.send(params)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"send (flatten + {len(plan)} collectives, world 1): {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.0f} GB/s staging rate")
    recv = BucketedReceiver(grp)
    recv._staging = sender._staging
    loaded = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = recv.receive(specs, lambda views: loaded.update(views))
    torch.cuda.synchronize()
    print(f"receive + unflatten views for {n} tensors: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    # reference protocol for comparison: one call per parameter
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _, p in params:
        grp.broadcast(p, src=0)
    torch.cuda.synchronize()
    print(f"per-tensor protocol call overhead ({len(params)} broadcasts, world 1): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    grp.close()


if __name__ == "__main__":
    main()
