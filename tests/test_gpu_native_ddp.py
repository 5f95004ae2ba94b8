"""NativeLearnerStep under data parallelism: two processes share cuda:0, form a gloo group, wrap the
same tiny LM in DistributedDataParallel and each run the HIP loss path on ITS share of two optimizer
steps (uneven shares: one rank has fewer micro-batches and fills up with sentinel passes).  Checked:
parameters identical on both ranks afterwards and equal to a single-process run over all the data
(DDP averages gradients: the single run uses lr / world), reduced statistics equal to the single-process
ones, `SamplesProcessed` published once per step, the weight-update interval rule, resume from
`TrainingMetrics` (reference finetune_loop.py:618-646, 698-713, 784-786, 805-808, 936-949)."""

from __future__ import annotations

import multiprocessing as mp
import os

import pytest

pytestmark = pytest.mark.gpu

V, SEQ, ATTEMPTS = 96, 40, 4
GROUPS_PER_STEP = 4  # 16 samples per optimizer step


def _model(torch):
    class TinyLM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(V, 24)
            self.out = torch.nn.Linear(24, V)

        def forward(self, input_ids=None, **kw):
            import types

            return types.SimpleNamespace(logits=self.out(torch.tanh(self.emb(input_ids))))

    torch.manual_seed(0)
    return TinyLM()


def _step_data(step: int):
    from pipelinerl_amd.synthetic import make_ragged

    rag, _ = make_ragged(GROUPS_PER_STEP, attempts=ATTEMPTS, seq_length=SEQ, vocab=V, seed=100 + step, prompt_min=3, prompt_max=8, with_ref=True)
    return rag


def _rl():
    from pipelinerl_amd.finetune.rl import RLConfig

    return RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05, divide_advantage_by_std=True,
                    clamp_log_ratio_ref_new_value=5)


def _worker(rank: int, world: int, port: int, tmp: str, out_q, backend: str = "gloo", own_gpu: bool = False) -> None:
    """`backend` "gloo" + one shared GPU (this file) or "nccl" + one GPU per rank (tests/test_gpu_multi.py)."""
    try:
        import torch
        import torch.distributed as dist

        from pipelinerl_amd import streams
        from pipelinerl_amd.finetune.types import TrainingMetrics
        from pipelinerl_amd.finetune_loop import TRAINER_TOPIC, NativeLearnerStep
        from pipelinerl_amd.hotpath import dense_micro_batches

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dev = torch.device("cuda", rank if own_gpu else 0)
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        streams.set_streams_backend("files")
        model = torch.nn.parallel.DistributedDataParallel(_model(torch).to(dev), device_ids=[dev.index] if backend == "nccl" else None)
        opt = torch.optim.SGD(model.parameters(), lr=0.2)
        sent = []

        class Manager:  # records what maybe_send_weights broadcasts
            def send_weight_update(self, version):
                sent.append(version)

        metrics = TrainingMetrics()
        metrics.samples, metrics.completed_steps, metrics.last_broadcasted_version = 160, 10, 160  # resumed run
        native = NativeLearnerStep(model, opt, _rl(), eos_token_id=2, samples_per_step=16, max_train_steps=20, process_group=dist.group.WORLD,
                                   training_metrics=metrics, weight_update_manager=Manager(), weight_update_interval=32,
                                   trainer_stream=streams.SingleStreamSpec(exp_path=tmp, topic=TRAINER_TOPIC))
        report = {"steps": []}
        for step in range(2):
            rag = _step_data(step)
            # uneven shares of whole groups: rank 0 takes 3 groups, rank 1 one
            lo, hi = (0, 3 * ATTEMPTS) if rank == 0 else (3 * ATTEMPTS, 4 * ATTEMPTS)
            mine = rag.select(range(lo, hi)).to(dev)
            res = native.step(mine, dense_micro_batches(mine, 2 * SEQ))
            broadcast = native.maybe_send_weights()
            report["steps"].append({"stats": native.stats_dict(res["stats"]), "micro_batches": res["micro_batches"],
                                    "sentinel_passes": res["sentinel_passes"], "broadcast": broadcast})
        native.finish()
        report["params"] = [p.detach().cpu().numpy() for p in model.module.parameters()]  # numpy: tensors would travel as fds of a dying process
        report["sent"] = sent
        report["metrics"] = (metrics.samples, metrics.completed_steps, metrics.last_broadcasted_version)
        dist.barrier()
        dist.destroy_process_group()
        out_q.put((rank, report))
    except Exception as e:  # noqa: BLE001
        import traceback

        out_q.put((rank, {"exception": f"{type(e).__name__}: {e}\n{traceback.format_exc()}"}))


def test_native_step_under_ddp_two_ranks(libprl, cuda_device, tmp_path):
    run_two_rank_check(cuda_device, tmp_path, backend="gloo", own_gpu=False)


def run_two_rank_check(cuda_device, tmp_path, backend: str, own_gpu: bool):
    import json
    import socket

    import torch

    from pipelinerl_amd.finetune.types import TrainingMetrics
    from pipelinerl_amd.finetune_loop import NativeLearnerStep
    from pipelinerl_amd.hotpath import dense_micro_batches

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q, backend, own_gpu), daemon=True) for r in range(2)]
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

    # single process over all the data; DDP averaged the two ranks' gradients, hence lr / 2
    model = _model(torch).to(cuda_device)
    metrics = TrainingMetrics()
    metrics.samples, metrics.completed_steps = 160, 10
    single = NativeLearnerStep(model, torch.optim.SGD(model.parameters(), lr=0.1), _rl(), eos_token_id=2, samples_per_step=16,
                               max_train_steps=20, training_metrics=metrics)
    want = []
    for step in range(2):
        rag = _step_data(step).to(cuda_device)
        res = single.step(rag, dense_micro_batches(rag, 2 * SEQ))
        want.append(single.stats_dict(res["stats"]))

    a, b = results[0], results[1]
    import numpy as np

    for pa, pb, ps in zip(a["params"], b["params"], model.parameters()):
        assert np.array_equal(pa, pb)  # the ranks stayed in lock-step
        np.testing.assert_allclose(pa, ps.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)
    for step in range(2):
        sa, sb = a["steps"][step], b["steps"][step]
        # the reduced statistics are the same on every rank; input_size is each rank's own token count
        assert {k: v for k, v in sa["stats"].items() if k != "input_size"} == {k: v for k, v in sb["stats"].items() if k != "input_size"}
        assert sa["stats"]["input_size"] + sb["stats"]["input_size"] == want[step]["input_size"]
        for k, w in want[step].items():
            if k != "input_size":
                assert abs(sa["stats"][k] - w) <= 1e-4 * max(abs(w), 1.0), (step, k, sa["stats"][k], w)
        assert sa["micro_batches"] > sb["micro_batches"] and sb["sentinel_passes"] == sa["micro_batches"] - sb["micro_batches"]
        assert sa["sentinel_passes"] == 0
    # resumed counters, one SamplesProcessed per step, weight-update interval of 32 samples = every second step
    assert a["metrics"] == (192, 12, 192) and a["sent"] == [192] and b["sent"] == [192]
    assert [s["broadcast"] for s in a["steps"]] == [False, True]
    lines = [json.loads(l) for l in next(tmp_path.rglob("*.jsonl")).read_text().splitlines()]
    assert [(l["kind"], l.get("samples_processed")) for l in lines] == [("samples_processed", 176), ("samples_processed", 192), ("training_done", None)]
