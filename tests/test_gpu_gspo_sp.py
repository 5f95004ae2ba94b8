"""GSPO under sequence parallelism (per-segment sums + SP all-reduce).

Two processes share cuda:0 and form a gloo group (the SP group); each runs this package's `rl_step`
on ITS slice of a packed batch and must reproduce what the reference produced on that slice in a
2-rank run of its own `rl_step(..., seq_parallel_group=group)` (tests/golden/make_gspo_sp_golden.py):
the full loss on every rank, the 32 statistics, d loss / d logits of the slice."""

from __future__ import annotations

import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = ["plain", "groupnorm_overlong"]
FP_TOL = 1e-4


def _worker(rank: int, world: int, port: int, out_q) -> None:
    try:
        import sys
        from pathlib import Path

        sys.path.insert(0, str(Path(__file__).resolve().parent))
        import torch
        import torch.distributed as dist

        from helpers import load_rl_case, rel_err
        from pipelinerl_amd.finetune.rl import RLConfig, rl_step
        from pipelinerl_amd.finetune.types import PipelineBatchEncoding

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)

        class FakeModel(torch.nn.Module):
            def __init__(self, logits):
                super().__init__()
                self.logits = torch.nn.Parameter(logits)

            def forward(self, **kw):
                import types

                return types.SimpleNamespace(logits=self.logits)

        report = {}
        for name in CASES:
            case = load_rl_case(f"gspo_sp2_{name}_rank{rank}")
            batch = PipelineBatchEncoding(**{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v)
                                             for k, v in case["batch"].items()}).to_device(dev)
            model = FakeModel(torch.from_numpy(case["logits"]).to(dev))
            cfg = RLConfig(**case["config"])
            loss, stats = rl_step(model, batch, *case["steps"], cfg, seq_parallel_group=dist.group.WORLD)
            loss.backward()
            errs = []
            if abs(loss.item() - case["loss"]) > FP_TOL * max(abs(case["loss"]), 1e-6) + 1e-7:
                errs.append(f"loss {loss.item()} vs {case['loss']}")
            if list(stats.keys()) != list(case["stats"].keys()):
                errs.append("stat keys differ")
            for k, w in case["stats"].items():
                if abs(float(stats[k]) - w) > FP_TOL * max(abs(w), 1.0):
                    errs.append(f"{k}: {stats[k]} vs {w}")
            grad = model.logits.grad.cpu().numpy()
            scale = np.abs(case["grad_logits"]).max()
            if scale == 0:
                if np.abs(grad).max() != 0:
                    errs.append("gradient should be zero")
            elif rel_err(grad, case["grad_logits"]) > FP_TOL:
                errs.append(f"grad rel err {rel_err(grad, case['grad_logits'])}")
            report[name] = errs
        dist.barrier()
        dist.destroy_process_group()
        out_q.put((rank, report))
    except Exception as e:  # noqa: BLE001
        import traceback

        out_q.put((rank, {"exception": [f"{type(e).__name__}: {e}", traceback.format_exc()]}))


def test_gspo_sequence_parallel_matches_reference(libprl, cuda_device):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket

    with socket.socket() as sock:  # a free rendezvous port
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
    for rank in (0, 1):
        for name, errs in results[rank].items():
            assert not errs, (rank, name, errs)


def _streamed_worker(rank: int, world: int, port: int, out_q) -> None:
    """The same two slices through `StreamedLearnerStep` (the model's own forward with the fused head, statistics left on the device):
    the SP group reaches the sequence-level sums through the model's forward."""
    try:
        import sys
        import types
        from pathlib import Path

        sys.path.insert(0, str(Path(__file__).resolve().parent))
        import torch
        import torch.distributed as dist

        from helpers import load_rl_case, rel_err
        from pipelinerl_amd.finetune.rl import RLConfig, host_stats, make_loss_config
        from pipelinerl_amd.finetune.types import PipelineBatchEncoding
        from pipelinerl_amd.finetune_loop import StreamedLearnerStep
        from pipelinerl_amd.fused_head import install_fused_head

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)

        class Body(torch.nn.Module):  # hidden states = the identity: the head's product IS the golden's logits (tests/test_gpu_fused_head_goldens.py)
            def __init__(self, T, H):
                super().__init__()
                h = torch.zeros(1, T, H, dtype=torch.bfloat16)
                h[0, torch.arange(T), torch.arange(T)] = 1.0
                self.register_buffer("h", h)

            def forward(self, input_ids=None, **kw):
                return (self.h,)

        class LM(torch.nn.Module):
            def __init__(self, logits):
                super().__init__()
                T, V = logits.shape
                H = -(-T // 64) * 64
                self.model = Body(T, H)
                self.lm_head = torch.nn.Linear(H, V, bias=False)
                with torch.no_grad():
                    self.lm_head.weight.zero_()
                    self.lm_head.weight[:, :T] = logits.t()

        report = {}
        for name in CASES:
            case = load_rl_case(f"gspo_sp2_{name}_rank{rank}")
            batch = PipelineBatchEncoding(**{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["batch"].items()}).to_device(dev)
            T = case["logits"].shape[1]
            lm = install_fused_head(LM(torch.from_numpy(case["logits"][0])).to(dev))
            cfg = RLConfig(**case["config"])
            # gradient_accumulation_passes = the golden's batch_size: the loss normaliser stays the golden's, and one micro-batch does not end the step
            step = StreamedLearnerStep(lm, torch.optim.SGD(lm.parameters(), lr=0.0), cfg, train_batch_size=1, gradient_accumulation_passes=cfg.batch_size,
                                       max_train_steps=case["steps"][1], seq_parallel=2, seq_parallel_group=dist.group.WORLD, send_weight_updates=False)
            step.metrics.completed_steps = case["steps"][0]
            res = step.step(batch)
            errs = []
            n_seq = len(case["batch"]["seq_boundaries"]) - 1 - (1 if case["batch"].get("padding") else 0)  # the SP filler is not a sample
            if res["did_optimizer_step"] or res["stats"] is not None or step.total_samples != n_seq:
                errs.append(f"accounting: {res['did_optimizer_step']}, {step.total_samples} samples for {n_seq} sequences on 2 SP ranks")
            loss = float(res["loss"].item())
            if abs(loss - case["loss"]) > 2 * FP_TOL * max(abs(case["loss"]), 1e-6) + 1e-7:
                errs.append(f"loss {loss} vs {case['loss']}")
            _, kl_coef, ent_coef = make_loss_config(cfg, *case["steps"])
            stats = host_stats(step._stats_dev[0], batch.input_ids.numel(), kl_coef, ent_coef)
            for k, w in case["stats"].items():
                if abs(float(stats[k]) - w) > 2 * FP_TOL * max(abs(w), 1.0):
                    errs.append(f"{k}: {stats[k]} vs {w}")
            grad = lm.lm_head.weight.grad[:, :T].t().float().cpu().numpy()  # d loss / d logits of this slice
            if rel_err(grad, case["grad_logits"][0]) > 2e-3:  # the reference's autograd on fp32 logits vs the head's two-bf16-plane weight
                errs.append(f"grad rel err {rel_err(grad, case['grad_logits'][0])}")
            report[name] = errs
        dist.barrier()
        dist.destroy_process_group()
        out_q.put((rank, report))
    except Exception as e:  # noqa: BLE001
        import traceback

        out_q.put((rank, {"exception": [f"{type(e).__name__}: {e}", traceback.format_exc()]}))


def test_streamed_learner_step_on_sequence_parallel_slices(libprl, cuda_device):
    """`StreamedLearnerStep(seq_parallel=2, seq_parallel_group=...)` on the reference's own 2-rank GSPO goldens: per-rank sample accounting
    (both ranks count the micro-batch's sequences, the total is halved, finetune_loop.py:709-713), the full loss on every rank, the
    statistics, and the slice's gradient."""
    import socket

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_streamed_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=420) for _ in range(2))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank in (0, 1):
        for name, errs in results[rank].items():
            assert not errs, (rank, name, errs)
