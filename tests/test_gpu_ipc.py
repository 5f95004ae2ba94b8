"""Colocated weight hand-off (trainer and inference worker on ONE GPU) over HIP IPC.

Reference behaviour being replaced: `pipelinerl/finetune_loop.py:205-292` + `pipelinerl/vllm1.py:83-134`
(the same NCCL broadcast is used even when both ends share a device).  Here the request carries
IPC handles of the trainer's buckets and the worker copies device-to-device.
"""

from __future__ import annotations

import multiprocessing as mp

import pytest
import torch

pytestmark = pytest.mark.gpu


def _param_set(seed: int, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = [("embed.weight", (1000, 64), torch.bfloat16), ("l0.w", (64, 64), torch.bfloat16),
              ("l0.b", (63,), torch.float32), ("l1.w", (129, 7), torch.bfloat16), ("norm", (1,), torch.float32)]
    return [(n, torch.randn(s, generator=g).to(dt).to(device)) for n, s, dt in shapes]


def _worker(req_q, ack_q):
    import torch

    from pipelinerl_amd.vllm_worker import StandaloneWeightReceiver

    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    module = torch.nn.Module()
    own = {}
    for n, p in _param_set(0, device):
        own[n] = torch.nn.Parameter(torch.zeros_like(p), requires_grad=False)
        module.register_parameter(n.replace(".", "__"), own[n])
    module.named_parameters = lambda: own.items()  # keep the dotted names
    recv = StandaloneWeightReceiver(module, device)
    ack_q.put("ready")
    while True:
        raw = req_q.get()
        if raw is None:
            break
        recv.receive_weight_update(raw)
        ack_q.put({n: t.float().double().sum().item() for n, t in own.items()})
    recv.close_communicator()


def test_colocated_ipc_weight_update(cuda_device):
    from pipelinerl_amd.finetune_loop import WeightUpdateManager

    ctx = mp.get_context("spawn")
    req_q, ack_q = ctx.Queue(), ctx.Queue()
    proc = ctx.Process(target=_worker, args=(req_q, ack_q), daemon=True)
    proc.start()
    try:
        assert ack_q.get(timeout=300) == "ready"
        acks = []

        def post(url, payload):
            assert url.endswith("/receive_weight_update")
            req_q.put(payload)
            acks.append(ack_q.get(timeout=120))

        params = _param_set(1, cuda_device)
        mgr = WeightUpdateManager(llm_urls=["ipc://worker"], accelerated_model=None, update_stream=None, actor_update_group=None,
                                  named_parameters_fn=lambda: params, transport="ipc", post=post)
        for version in (1, 2):
            if version == 2:  # in-place optimizer step; same buckets and handles are reused
                for _, p in params:
                    p.mul_(0.5).add_(1.0)
            mgr.send_weight_update(version)
            want = {n: p.float().double().sum().item() for n, p in params}
            assert acks[-1] == pytest.approx(want, rel=0, abs=0)
        mgr.shutdown()
    finally:
        req_q.put(None)
        proc.join(timeout=60)
        if proc.is_alive():
            proc.kill()


def _cut_worker(req_q, ack_q, bucket_bytes, max_allocation):
    import torch

    from pipelinerl_amd.weight_sync import ColocatedReceiver, ParamSpec

    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    # max_allocation None: the receiver is built like vllm_worker builds it (bucket_bytes only) and takes the sender's cap from the request
    rx = ColocatedReceiver(device, bucket_bytes, max_allocation) if max_allocation is not None else ColocatedReceiver(device, bucket_bytes)
    ack_q.put("ready")
    while True:
        msg = req_q.get()
        if msg is None:
            break
        info = [ParamSpec(n, tuple(s), getattr(torch, dt)) for n, s, dt in msg["info"]]
        # `big.direct` has a registered destination (its row ranges are scattered straight into it), `big.loaded` comes back whole
        dest = {n: torch.zeros(tuple(s), dtype=getattr(torch, dt), device=device) for n, s, dt in msg["info"] if n in ("big.direct", "small")}
        loaded = {}
        n = rx.receive(info, msg["handles"], msg["nbytes"], lambda views: loaded.update({k: v.clone() for k, v in views}), dest,
                       max_allocation=msg.get("max_allocation"))
        out = {k: v.double().sum().item() for k, v in {**dest, **loaded}.items()}
        ack_q.put((n, out, sorted(loaded)))
    rx.close()


@pytest.mark.parametrize("cap_from", ["constructor", "request"])
def test_ipc_hand_off_of_tensors_cut_into_row_ranges(cuda_device, cap_from):
    """(`cap_from` = request: the receiver knows only `bucket_bytes`, the allocation cap that fixes the piece list travels with the
    update - `WeightUpdateRequest.ipc_max_allocation`.)  Tensors too large for one exportable allocation (`weight_sync.IPC_MAX_ALLOCATION`; small limits here) cross in row ranges:
    the sender keeps such a tensor where it is and copies its ranges per update, every other parameter is rehomed; the receiver
    scatters ranges into a registered destination or reassembles the tensor for `load_weights`."""
    from pipelinerl_amd.weight_sync import ColocatedSender

    bucket_bytes, max_allocation = 4096, 8192
    g = torch.Generator(device="cpu").manual_seed(3)
    params = [(n, torch.nn.Parameter(torch.randn(s, generator=g).to(dt).to(cuda_device), requires_grad=False)) for n, s, dt in
              (("small", (7, 9), torch.float32), ("big.direct", (300, 16), torch.float32), ("mid", (40, 8), torch.bfloat16), ("big.loaded", (257, 24), torch.bfloat16))]
    ctx = mp.get_context("spawn")
    req_q, ack_q = ctx.Queue(), ctx.Queue()
    proc = ctx.Process(target=_cut_worker, args=(req_q, ack_q, bucket_bytes, max_allocation if cap_from == "constructor" else None), daemon=True)
    proc.start()
    tx = ColocatedSender(cuda_device, bucket_bytes, max_allocation)
    try:
        assert ack_q.get(timeout=300) == "ready"
        before = {n: p.data_ptr() for n, p in params}
        tx.rehome(params)
        moved = {n for n, p in params if p.data_ptr() != before[n]}
        assert moved == {"small", "mid"}, "the two tensors that fit were rehomed, the cut ones stay where they are"
        for version in range(2):
            if version:
                for _, p in params:
                    p.data.mul_(0.5).add_(1.0)
            pub = tx.publish([(n, p.data) for n, p in params])
            assert max(pub["ipc_nbytes"]) < max_allocation and len(pub["ipc_handles"]) > 4
            req_q.put({"info": [(n, list(p.shape), str(p.dtype).replace("torch.", "")) for n, p in params], "handles": pub["ipc_handles"], "nbytes": pub["ipc_nbytes"],
                       "max_allocation": pub["ipc_max_allocation"] if cap_from == "request" else None})
            n, sums, loaded = ack_q.get(timeout=120)
            assert n == 4 and loaded == ["big.loaded", "mid"]
            assert sums == pytest.approx({k: p.data.double().sum().item() for k, p in params}, rel=0, abs=0)
    finally:
        req_q.put(None)
        proc.join(timeout=60)
        if proc.is_alive():
            proc.kill()
        tx.close()
