"""Measurement helper: trainer -> inference-worker weight hand-off with both processes on ONE GPU
(`transport="ipc"`, BASELINE.json configs[1]'s colocated layout).  Used by `bench.py` at N = 1 and by
`scripts/wsync_colocated_bench.py`; not part of the product path.

The protocol exercised is the reference's (finetune_loop.py:205-292 -> vllm1.py:83-134): the
trainer announces the parameter list in a blocking request, the worker loads every tensor and
acknowledges; only the byte transport differs (HIP IPC buckets instead of an NCCL broadcast).
"""

from __future__ import annotations

import multiprocessing as mp
import time

import torch


def qwen25_shapes(which: str):
    # (hidden, intermediate, vocab, layers, kv_heads * head_dim, tied embeddings) of the BASELINE.json model sizes; "32b" =
    # configs[4]: 771 tensors, 32.76 G parameters = 65.5 GB in bf16 (finetune_loop.py:205-292 broadcasts each of them)
    H, I, V, L, KV, tied = {"7b": (3584, 18944, 152064, 28, 512, False), "0p5b": (896, 4864, 151936, 24, 128, True),
                            "32b": (5120, 27648, 152064, 64, 1024, False)}[which]
    out = [("model.embed_tokens.weight", (V, H))]
    for i in range(L):
        p = f"model.layers.{i}."
        out += [(p + "self_attn.q_proj.weight", (H, H)), (p + "self_attn.q_proj.bias", (H,)), (p + "self_attn.k_proj.weight", (KV, H)),
                (p + "self_attn.k_proj.bias", (KV,)), (p + "self_attn.v_proj.weight", (KV, H)), (p + "self_attn.v_proj.bias", (KV,)),
                (p + "self_attn.o_proj.weight", (H, H)), (p + "mlp.gate_proj.weight", (I, H)), (p + "mlp.up_proj.weight", (I, H)),
                (p + "mlp.down_proj.weight", (H, I)), (p + "input_layernorm.weight", (H,)), (p + "post_attention_layernorm.weight", (H,))]
    out += [("model.norm.weight", (H,))]
    if not tied:
        out += [("lm_head.weight", (V, H))]
    return out


def _worker(which, req_q, ack_q, direct=True):
    import torch

    from pipelinerl_amd.vllm_worker import WorkerExtension

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    own = {n: torch.zeros(s, dtype=torch.bfloat16, device=dev) for n, s in qwen25_shapes(which)}

    class Engine(WorkerExtension):
        device, rank = dev, 0

        def _weight_destinations(self):
            return own if direct else None

        def _load_weights(self, weights):
            names = [n for n, _ in weights]
            torch._foreach_copy_([own[n] for n in names], [t for _, t in weights])
            return names

    eng = Engine()
    ack_q.put("ready")
    while True:
        req = req_q.get()
        if req is None:
            break
        t0 = time.perf_counter()
        eng.receive_weight_update(req)
        ack_q.put({"recv_ms": (time.perf_counter() - t0) * 1e3, "probe": own["model.norm.weight"][:4].float().tolist()})
    eng.close_communicator()


def _get(q, proc, timeout: float):
    """`q.get` that gives up early when the worker process has died."""
    import queue

    deadline = time.monotonic() + timeout
    while True:
        try:
            return q.get(timeout=1.0)
        except queue.Empty:
            if not proc.is_alive():
                raise RuntimeError(f"weight-sync worker exited with code {proc.exitcode}") from None
            if time.monotonic() > deadline:
                raise TimeoutError("weight-sync worker did not answer in time") from None


def colocated_probe(which: str = "7b", iters: int = 5, rehome: bool = True, ready_timeout: float = 600.0, direct: bool = True) -> dict:
    """Median request-to-ack time of `WeightUpdateManager.send_weight_update` over `iters` updates of
    the Qwen2.5 `which` parameter set (bf16), the worker being a second process on cuda:0."""
    from pipelinerl_amd.finetune_loop import WeightUpdateManager
    from pipelinerl_amd.weight_sync import ColocatedSender

    dev = torch.device("cuda", 0)
    ctx = mp.get_context("spawn")
    req_q, ack_q = ctx.Queue(), ctx.Queue()
    proc = ctx.Process(target=_worker, args=(which, req_q, ack_q, direct), daemon=True)
    proc.start()
    mgr = None
    try:
        params = [(n, torch.nn.Parameter(torch.empty(s, dtype=torch.bfloat16, device=dev).normal_(), requires_grad=False))
                  for n, s in qwen25_shapes(which)]
        nbytes = sum(p.numel() * 2 for _, p in params)
        if _get(ack_q, proc, ready_timeout) != "ready":
            raise RuntimeError("unexpected first message from the weight-sync worker")
        acks = []

        def post(url, payload):
            req_q.put(payload)
            acks.append(_get(ack_q, proc, 300.0))

        mgr = WeightUpdateManager(llm_urls=["ipc://worker"], accelerated_model=None, update_stream=None, actor_update_group=None,
                                  named_parameters_fn=lambda: params, transport="ipc", post=post)
        if rehome:
            mgr._sender = ColocatedSender(dev, mgr.bucket_bytes)
            mgr._sender.rehome(params)
        times = []
        for it in range(iters + 1):
            for _, p in params[-2:]:
                p.data.add_(1.0)  # the "optimizer step"
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mgr.send_weight_update(it + 1)
            times.append((time.perf_counter() - t0) * 1e3)
            if len(acks) != it + 1:
                raise RuntimeError("the worker did not acknowledge the update")
            want = dict(params)["model.norm.weight"][:4].float().tolist()
            if acks[-1]["probe"] != want:
                raise RuntimeError(f"worker weights differ after the update: {acks[-1]['probe']} != {want}")
        steady = sorted(times[1:])
        recv = sorted(a["recv_ms"] for a in acks[1:])
        med, rmed = steady[len(steady) // 2], recv[len(recv) // 2]
        return {"metric": "trainer_to_actor_weight_sync_ms", "layout": "colocated (1 GPU, 2 processes, HIP IPC)", "params": which,
                "tensors": len(params), "gbytes": round(nbytes / 1e9, 3), "zero_copy_publish": rehome, "worker_scatter_kernel": direct,
                "first_ms": round(times[0], 2), "median_ms": round(med, 2), "min_ms": round(steady[0], 2),
                "worker_copy_ms": round(rmed, 2), "effective_GBps": round(nbytes / med / 1e6, 1)}
    finally:
        req_q.put(None)  # the worker unmaps the buckets before they are freed
        proc.join(timeout=60)
        if proc.is_alive():
            proc.kill()
        if mgr is not None:
            mgr.shutdown()
            if mgr._sender is not None:
                params = None
                mgr._sender.close()
