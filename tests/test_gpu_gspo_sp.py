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
