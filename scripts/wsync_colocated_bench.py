"""Trainer -> inference-worker weight hand-off when both share ONE MI355X (BASELINE.json configs[2]'s
colocated layout), over HIP IPC: two processes, the request carries the bucket handles, the worker
copies device-to-device into its own weights.  Prints one JSON line per parameter set.

    python scripts/wsync_colocated_bench.py [--sets 0p5b,7b] [--iters 5]
"""

import argparse
import json
import multiprocessing as mp
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def qwen25_shapes(which: str):
    H, I, V, L, KV, tied = {"7b": (3584, 18944, 152064, 28, 512, False), "0p5b": (896, 4864, 151936, 24, 128, True)}[which]
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


def worker(which, req_q, ack_q):
    import torch

    from pipelinerl_amd.vllm_worker import WorkerExtension

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    own = {n: torch.zeros(s, dtype=torch.bfloat16, device=dev) for n, s in qwen25_shapes(which)}

    class Engine(WorkerExtension):
        device, rank = dev, 0

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


def run(which: str, iters: int, rehome: bool):
    from pipelinerl_amd.finetune_loop import WeightUpdateManager

    dev = torch.device("cuda", 0)
    ctx = mp.get_context("spawn")
    req_q, ack_q = ctx.Queue(), ctx.Queue()
    proc = ctx.Process(target=worker, args=(which, req_q, ack_q), daemon=True)
    proc.start()
    params = [(n, torch.nn.Parameter(torch.empty(s, dtype=torch.bfloat16, device=dev).normal_(), requires_grad=False))
              for n, s in qwen25_shapes(which)]
    nbytes = sum(p.numel() * 2 for _, p in params)
    assert ack_q.get(timeout=600) == "ready"
    acks = []

    def post(url, payload):
        req_q.put(payload)
        acks.append(ack_q.get(timeout=300))

    mgr = WeightUpdateManager(llm_urls=["ipc://worker"], accelerated_model=None, update_stream=None, actor_update_group=None,
                              named_parameters_fn=lambda: params, transport="ipc", post=post)
    if rehome:
        from pipelinerl_amd.weight_sync import ColocatedSender

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
        want = dict(params)["model.norm.weight"][:4].float().tolist()
        assert acks[-1]["probe"] == want, (acks[-1], want)
    mgr.shutdown()
    req_q.put(None)
    proc.join(timeout=60)
    steady = sorted(times[1:])
    recv = sorted(a["recv_ms"] for a in acks[1:])
    med, rmed = steady[len(steady) // 2], recv[len(recv) // 2]
    return {"metric": "trainer_to_actor_weight_sync_ms", "layout": "colocated (1 GPU, 2 processes, HIP IPC)", "params": which,
            "tensors": len(params), "gbytes": round(nbytes / 1e9, 3), "zero_copy_publish": rehome,
            "first_ms": round(times[0], 2), "median_ms": round(med, 2), "min_ms": round(steady[0], 2),
            "worker_copy_ms": round(rmed, 2), "effective_GBps": round(nbytes / med / 1e6, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="0p5b,7b")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    for which in args.sets.split(","):
        for rehome in (False, True):
            print(json.dumps(run(which, args.iters, rehome)), flush=True)


if __name__ == "__main__":
    main()
