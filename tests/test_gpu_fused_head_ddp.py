"""The fused output head under data parallelism: two processes share cuda:0, form a gloo group, prepare
the same tiny body + lm_head model with `install_fused_head`, wrap it in DistributedDataParallel and drive
it with `rl_step_fused_head` - the loss is produced INSIDE the wrapper's forward, so DDP arms its gradient
reduction for the body's parameters and for the head weight (whose gradient comes from the hand-written
backward).  Two optimizer steps of two micro-batches each, the first under `no_sync()`.  Checked: parameters
identical on both ranks and equal to a single-process run over all four micro-batches per step with the
gradients scaled by 1 / world (reference finetune_loop.py:698-713, 768-808: DDP / accelerate own the
reduction, the loss function only has to be called through the wrapped model)."""

from __future__ import annotations

import multiprocessing as mp
import os

import pytest

pytestmark = pytest.mark.gpu

V, H, T = 1024, 128, 96
CFG = dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05, entropy_bonus=0.01,
           final_entropy_bonus=0.01, temperature=0.7, batch_size=8, clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False)


def _model(torch):
    import types

    class Body(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(V, H)
            self.lin = torch.nn.Linear(H, H)

        def forward(self, input_ids=None, **kw):
            return (torch.tanh(self.lin(self.emb(input_ids))).to(torch.bfloat16),)

    class LM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Body()
            self.lm_head = torch.nn.Linear(H, V, bias=False)

        def forward(self, **kw):
            h = self.model(**kw)[0]
            return types.SimpleNamespace(logits=h.float() @ self.lm_head.weight.t())

    torch.manual_seed(0)
    return LM()


def _batch(torch, model, seed: int, dev):
    """A packed micro-batch of two sequences whose old log-probs sit near the model's own (inside the clip range)."""
    import numpy as np

    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    rng = np.random.default_rng(seed)
    ids = rng.integers(0, V, size=(1, T), dtype=np.int64)
    labels = ids.copy()
    labels[0, :8] = -100
    half = T // 2
    labels[0, half : half + 6] = -100
    pos = np.concatenate([np.arange(half), np.arange(T - half)])[None].astype(np.int64)
    with torch.no_grad():
        t_ids = torch.from_numpy(ids).to(dev)
        logits = model(input_ids=t_ids).logits[0].double() / CFG["temperature"]
        lp = torch.log_softmax(logits, -1)
        nlp = np.concatenate([[0.0], lp[torch.arange(T - 1), t_ids[0, 1:]].cpu().numpy()])
    old = nlp + rng.normal(0, 0.05, T)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)[None])  # noqa: E731
    n_lab = float((labels != -100).sum())
    return PipelineBatchEncoding(
        input_ids=torch.from_numpy(ids), labels=torch.from_numpy(labels), position_ids=torch.from_numpy(pos),
        attention_mask=torch.ones(1, T, dtype=torch.int64), old_logprobs=f32(old), ref_logprobs=f32(old + rng.normal(0, 0.05, T)),
        advantages=f32(rng.normal(0, 1, T)), rewards=f32(rng.integers(0, 2, T)), group_tokens=f32(np.full(T, 31.0)),
        num_labels=f32(np.full(T, n_lab)), overflow=f32(np.zeros(T)), model_version=0, is_packed=True).to_device(dev)


def _seeds(step: int, rank: int) -> list[int]:
    return [1000 * step + 10 * rank + k for k in range(2)]


def _worker(rank: int, world: int, port: int, out_q) -> None:
    try:
        import torch
        import torch.distributed as dist

        from pipelinerl_amd.finetune.rl import RLConfig
        from pipelinerl_amd.fused_head import install_fused_head, rl_step_fused_head

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        lm = install_fused_head(_model(torch).to(dev))
        names = [n for n, _ in lm.named_parameters()]
        ddp = torch.nn.parallel.DistributedDataParallel(lm)
        opt = torch.optim.SGD(ddp.parameters(), lr=0.5)
        cfg = RLConfig(**CFG)
        report = {"loss": [], "keys": None, "names": names}
        for step in range(2):
            batches = [_batch(torch, lm, s, dev) for s in _seeds(step, rank)]
            opt.zero_grad(set_to_none=True)
            with ddp.no_sync():
                loss, stats = rl_step_fused_head(ddp, batches[0], 2, 10, cfg)
                loss.backward()
            report["loss"].append(loss.item())
            loss, stats = rl_step_fused_head(ddp, batches[1], 2, 10, cfg)
            loss.backward()
            report["loss"].append(loss.item())
            report["keys"] = list(stats)
            opt.step()
        report["params"] = [p.detach().cpu().numpy() for p in lm.parameters()]
        # the ordinary forward still works on the prepared model
        report["logits_shape"] = tuple(lm(input_ids=batches[0].input_ids).logits.shape)
        dist.barrier()
        dist.destroy_process_group()
        out_q.put((rank, report))
    except Exception as e:  # noqa: BLE001
        import traceback

        out_q.put((rank, {"exception": f"{type(e).__name__}: {e}\n{traceback.format_exc()}"}))


def test_fused_head_under_ddp_two_ranks(libprl, cuda_device):
    import socket

    import numpy as np
    import torch

    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.fused_head import rl_step_fused_head

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=420) for _ in range(2))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for r in (0, 1):
        assert "exception" not in results[r], results[r]["exception"]

    # single process, bare model, all four micro-batches of a step; DDP averaged the ranks' summed gradients
    lm = _model(torch).to(cuda_device)
    opt = torch.optim.SGD(lm.parameters(), lr=0.5)
    cfg = RLConfig(**CFG)
    want_loss = {0: [], 1: []}
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        batches = {r: [_batch(torch, lm, s, cuda_device) for s in _seeds(step, r)] for r in (0, 1)}
        for r in (0, 1):
            for b in batches[r]:
                loss, stats = rl_step_fused_head(lm, b, 2, 10, cfg)
                (loss * 0.5).backward()
                want_loss[r].append(loss.item())
        opt.step()

    a, b = results[0], results[1]
    assert a["names"] == [n for n, _ in lm.named_parameters()]  # install_fused_head renames nothing
    assert a["logits_shape"] == (1, T, V)
    assert a["keys"] == list(stats)
    init = [p.detach().cpu().numpy() for p in _model(torch).parameters()]
    for pa, pb, ps, p0 in zip(a["params"], b["params"], lm.parameters(), init):
        assert np.array_equal(pa, pb)  # the ranks stayed in lock-step
        # step 2 starts from parameters that differ in the last fp32 bits, and a bf16 hidden state on a rounding
        # boundary then moves by a whole bf16 ulp: a handful of elements differ by ~5e-5.  A missing reduction
        # (one rank's gradient instead of the average) would show up at the size of the update itself:
        moved = np.abs(pa - p0).max()
        np.testing.assert_allclose(pa, ps.detach().cpu().numpy(), rtol=1e-3, atol=min(1e-4, 0.05 * moved))
    assert np.abs(a["params"][-1] - init[-1]).max() > 2e-3  # the head weight moved by far more than the tolerance
    for r in (0, 1):
        np.testing.assert_allclose(results[r]["loss"], want_loss[r], rtol=1e-4)
