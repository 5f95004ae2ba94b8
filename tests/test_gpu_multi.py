"""Paths that need MORE THAN ONE GPU (skipped on the 1-GPU development boxes, run wherever >= 2 devices are
visible): the RCCL weight broadcast that has to cross a link, and the learner step under DDP over RCCL.

  * `WeightSyncGroup.from_init_method("tcp://...")`: the reference's rendezvous (torch_utils.py:70-94), one
    rank per GPU, rank 0 = trainer (vllm1.py:71);
  * `broadcast_bucket` in both modes (ncclBroadcast; scatter + all-gather over the pairwise xGMI links)
    byte-exact for awkward sizes: 1 byte, 255, 4099, 1 GiB + 3;
  * `BucketedSender` / `BucketedReceiver`: the whole Qwen2.5-7B parameter set (339 tensors, 15.2 GB) through
    1 GiB buckets and the two-stream pipeline, EVERY tensor compared on every receiver
    (reference finetune_loop.py:205-292, vllm1.py:110-127);
  * the TP-aware update (`WeightSyncGroup.tp_shard_groups`, `ShardedSender`): one communicator per tensor-parallel
    rank, every worker receives only its slices of the 7B parameter set (needs 3 / 5 GPUs);
  * `NativeLearnerStep` under DistributedDataParallel over RCCL with the HIP loss (same check as
    tests/test_gpu_native_ddp.py, which runs it with two processes on one GPU over gloo)."""

from __future__ import annotations

import multiprocessing as mp
import socket
import time
from pathlib import Path

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least 2 GPUs")]

SIZES = [1, 255, 4099, (1 << 30) + 3]
DIAG = Path(__file__).resolve().parent.parent / "gpurun_out" / "multi_gpu_diag.jsonl"


def _diag(record: dict) -> None:
    """The first run of these tests on a multi-GPU node is also the first time the code moves bytes between two devices: every test leaves
    what it did - transport, world size, bytes, seconds, bus bandwidth, which RCCL path - on stdout (pytest -rA / -s) and in
    gpurun_out/multi_gpu_diag.jsonl, so that a failure (or a slow link) can be read without re-running."""
    import json

    line = json.dumps(record)
    print("[multi-gpu diag]", line, flush=True)
    try:
        DIAG.parent.mkdir(parents=True, exist_ok=True)
        with open(DIAG, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _env_diag() -> dict:
    import os

    props = [torch.cuda.get_device_properties(i) for i in range(torch.cuda.device_count())]
    return {"devices": len(props), "names": sorted({p.name for p in props}), "gcn": sorted({getattr(p, "gcnArchName", "?") for p in props}),
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "NCCL_DEBUG": os.environ.get("NCCL_DEBUG"),
            "torch": torch.__version__, "hip": torch.version.hip}


def _free_port() -> int:
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _pattern(nbytes: int, seed: int, dev) -> "torch.Tensor":
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev, generator=g)


def _wsync_worker(rank: int, world: int, port: int, out_q, which: str = "7b") -> None:
    try:
        from pipelinerl_amd.weight_sync import BucketedReceiver, BucketedSender, ParamSpec, WeightSyncGroup
        from pipelinerl_amd.weight_sync_probe import qwen25_shapes

        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        grp = WeightSyncGroup.from_init_method(f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device=dev, timeout_s=120)
        errs = []
        diag = {"rank": rank, "device": torch.cuda.get_device_name(dev), "rccl_comm": list(grp.comm_size()), "buckets": {}}
        # ---- raw buckets, both modes, awkward sizes
        for mode in ("broadcast", "scatter_allgather"):
            for k, n in enumerate(SIZES):
                want = _pattern(n, 1000 + k, dev)
                buf = want.clone() if rank == 0 else torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                grp.broadcast_bucket(buf, mode=mode)
                torch.cuda.synchronize()
                diag["buckets"][f"{mode}:{n}"] = round(1e3 * (time.perf_counter() - t0), 3)
                if not torch.equal(buf, want):
                    bad = int((buf != want).sum())
                    errs.append(f"{mode} {n} bytes: {bad} bytes differ on rank {rank}")
                del buf, want
        # ---- the whole update (339 tensors / 15.2 GB for 7B, 771 tensors / 65.5 GB for 32B) through the bucketed sender /
        # receiver, every tensor verified
        shapes = qwen25_shapes(which)
        gen = torch.Generator(device=dev).manual_seed(77)
        params = [(n, torch.empty(s, dtype=torch.bfloat16, device=dev).normal_(generator=gen)) for n, s in shapes]
        nbytes = sum(t.numel() * 2 for _, t in params)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if rank == 0:
            BucketedSender(grp, 1 << 30).send(params)
            torch.cuda.synchronize()
        else:
            dest = {n: torch.zeros_like(t) for n, t in params}
            info = [ParamSpec(n, tuple(s), torch.bfloat16) for n, s in shapes]
            t0 = time.perf_counter()
            got = BucketedReceiver(grp, 1 << 30).receive(info, None, destinations=dest)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # first update of a fresh communicator (channel set-up included): a lower bound on the steady-state rate, which bench.py --gpus N measures
        diag["full_update"] = {"tensors": len(shapes), "gbytes": round(nbytes / 1e9, 3), "ms_first": round(1e3 * dt, 1), "GBps_first": round(nbytes / dt / 1e9, 1),
                               "mode": "scatter_allgather (default)", "per_link_peak_GBps": 153}
        if rank != 0:
            if got != len(shapes):
                errs.append(f"received {got} of {len(shapes)} tensors")
            wrong = [n for n, t in params if not torch.equal(dest[n], t)]
            if wrong:
                errs.append(f"{len(wrong)} tensors differ, first {wrong[:3]}")
        torch.cuda.synchronize()
        grp.close()
        out_q.put((rank, errs, diag))
    except Exception as e:  # noqa: BLE001
        import traceback

        out_q.put((rank, [f"{type(e).__name__}: {e}", traceback.format_exc()], {"rank": rank, "died": True}))


def _collect(procs, q, world, timeout=600):
    """(errors per rank, diagnostics per rank); a rank that never reports is named instead of a bare queue.Empty."""
    import queue

    for p in procs:
        p.start()
    got = {}
    try:
        deadline = time.time() + timeout
        while len(got) < world and time.time() < deadline:
            try:
                r = q.get(timeout=5)
                got[r[0]] = r[1:]
            except queue.Empty:
                if all(not p.is_alive() for p in procs):
                    break
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    missing = [r for r in range(world) if r not in got]
    errs = {r: got[r][0] for r in got}
    for r in missing:
        errs[r] = [f"rank {r} never reported (exit code {procs[r].exitcode}): a hang in the rendezvous or in a collective - rerun with NCCL_DEBUG=INFO"]
    return errs, {r: (got[r][1] if len(got[r]) > 1 else None) for r in got}


@pytest.mark.parametrize("world,which", [(2, "7b"), (4, "7b"), (8, "7b"), (3, "32b")], ids=["2_gpus_7b", "4_gpus_7b", "8_gpus_7b", "3_gpus_32b_65GB"])
def test_rccl_weight_broadcast_is_byte_exact(libprl, world, which):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wsync_worker, args=(r, world, port, q, which), daemon=True) for r in range(world)]
    results, diags = _collect(procs, q, world)
    _diag({"test": "rccl_weight_broadcast", "transport": "rccl_xgmi", "world": world, "params": which, "env": _env_diag(), "ranks": diags})
    for r in range(world):
        assert not results[r], (r, results[r])


def _tp_shard_worker(rank: int, world: int, port: int, tp: int, out_q, which: str = "7b") -> None:
    """rank 0 = trainer; ranks 1.. = TP rank (r - 1) % tp of engine (r - 1) // tp, one GPU each."""
    try:
        from pipelinerl_amd.tp_shard import plan_tp_shards, shard_view
        from pipelinerl_amd.vllm_worker import StandaloneShardReceiver
        from pipelinerl_amd.finetune_loop import ParameterInfo, WeightUpdateRequest
        from pipelinerl_amd.weight_sync import ShardedSender, WeightSyncGroup
        from pipelinerl_amd.weight_sync_probe import qwen25_shapes

        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        groups = WeightSyncGroup.tp_shard_groups(f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, tp_size=tp, device=dev, timeout_s=120)
        kv_heads = {"7b": 4, "32b": 8}[which]
        shapes = qwen25_shapes(which)
        cuts = plan_tp_shards(shapes, tp, kv_heads=kv_heads)
        gen = torch.Generator(device=dev).manual_seed(91)  # the same tensors on every rank: the workers check against them
        params = [(n, torch.empty(s, dtype=torch.bfloat16, device=dev).normal_(generator=gen)) for n, s in shapes]
        errs = []
        if rank == 0:
            sender = ShardedSender(groups, 1 << 30)
            sender.send(params, cuts)
            torch.cuda.synchronize()
            total = sum(t.numel() * 2 for _, t in params)
            if not all(b < 0.52 * total for b in sender.bytes_sent):
                errs.append(f"bytes per TP rank {sender.bytes_sent} of {total}")
        else:
            t = (rank - 1) % tp
            w = StandaloneShardReceiver(shapes, lambda n: torch.bfloat16, dev, t, tp, kv_heads=kv_heads)
            w.model_update_group, w.tp_rank, w.tp_size = groups[0], t, tp
            req = WeightUpdateRequest(version=1, transport="sharded", bucket_bytes=1 << 30, tp_size=tp,
                                      parameters_info=[ParameterInfo(name=n, shape=list(s), dtype="torch.bfloat16", shard_dim=cuts[n].dim,
                                                                     shard_parts=cuts[n].parts) for n, s in shapes])
            w.receive_weight_update(req.model_dump_json())
            torch.cuda.synchronize()
            wrong = [n for n, x in params if not torch.equal(w.slices[n], shard_view(x, cuts[n], t, tp))]
            if wrong:
                errs.append(f"{len(wrong)} slices differ on rank {rank} (TP rank {t}), first {wrong[:3]}")
        for g in groups:
            g.close()
        out_q.put((rank, errs, {"rank": rank, "role": "trainer" if rank == 0 else f"engine {(rank - 1) // tp} tp {(rank - 1) % tp}",
                                "bytes_sent_per_tp_rank": getattr(locals().get("sender"), "bytes_sent", None)}))
    except Exception as e:  # noqa: BLE001
        import traceback

        out_q.put((rank, [f"{type(e).__name__}: {e}", traceback.format_exc()]))


@pytest.mark.parametrize("engines,tp,which", [(1, 2, "7b"), (2, 2, "7b"), (2, 2, "32b")],
                         ids=["1_engine_tp2_7b", "2_engines_tp2_7b", "configs4_2_engines_tp2_32b"])
def test_tp_aware_update_over_rccl(libprl, engines, tp, which):
    """The TP-aware update over RCCL: one communicator per TP rank, every worker receives half of the parameter set
    (its own slices), verified slice by slice.  `configs4_...`: BASELINE.json configs[4]'s receiver layout - two TP = 2
    engines and the trainer, the 771-tensor / 65.5 GB Qwen2.5-32B set, 8 KV heads."""
    world = 1 + engines * tp
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_shard_worker, args=(r, world, port, tp, q, which), daemon=True) for r in range(world)]
    results, diags = _collect(procs, q, world)
    _diag({"test": "tp_aware_update", "transport": "rccl_xgmi, one communicator per TP rank", "world": world, "engines": engines, "tp": tp, "params": which,
           "env": _env_diag(), "ranks": diags})
    for r in range(world):
        assert not results[r], (r, results[r])


def test_native_learner_step_under_ddp_over_rccl(libprl, cuda_device, tmp_path):
    from test_gpu_native_ddp import run_two_rank_check

    run_two_rank_check(cuda_device, tmp_path, backend="nccl", own_gpu=True)


@pytest.mark.parametrize("n_learners,n_engines,tp", [(1, 1, 1), (2, 2, 1), (4, 4, 1), (1, 1, 2), (4, 2, 2)],
                         ids=["1+1", "configs2_2+2", "configs3_4+4", "1+1xTP2", "configs4_4+2xTP2"])
def test_pipeline_learners_and_engines_on_their_own_gpus_over_rccl(libprl, tmp_path, n_learners, n_engines, tp):
    """BASELINE configs[2] / [3] as they are meant to run: every engine and every learner rank on its OWN GPU (engines first, world.py:143-192),
    gradients over RCCL, trainer rank 0 -> the weight-update group of M + 1 over RCCL / xGMI after every optimizer step.  The same
    `run_pipeline` the 1-GPU tests drive with gloo / HIP IPC (tests/test_gpu_pipeline_procs.py, tests/test_pipeline_topology_cpu.py)."""
    import json

    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    need = n_learners + n_engines * tp
    if torch.cuda.device_count() < need:
        pytest.skip(f"needs {need} GPUs")
    bs, steps = 32, 3
    spec = PipelineSpec(exp_path=str(tmp_path / "exp"), model="tiny", global_batch=bs, seq_length=128, attempts=4, steps=steps, n_problems=5, concurrent_groups=2,
                        stage_timeout_s=600.0, n_learners=n_learners, n_engines=n_engines, engine_tp=tp, weight_transport="rccl", share_device=False, stacks_after_s=300.0,
                        kl_coef=0.001 if tp > 1 else 0.0)
    res = run_pipeline(spec)
    s = res.get("summary") or {}
    _diag({"test": "pipeline_rccl", "learners": n_learners, "engines": n_engines, "engine_tp": tp, "env": _env_diag(), "error": res.get("error"),
           "topology": s.get("topology"), "weight_sync_under_load_ms": s.get("weight_sync_under_load_ms"), "samples_per_s": s.get("samples_per_s")})
    assert "error" not in res, json.dumps(res.get("error"), indent=1)[:6000]
    st = res["stages"]
    eng = [n for n in st if n.startswith("engine")]
    lrn = [n for n in st if n.startswith("learner")]
    assert len({st[n]["device"] for n in eng + lrn}) == n_learners + n_engines, "one (first) GPU per engine and per learner rank"
    assert all(st[n]["completed_steps"] == steps and st[n]["local_samples"] == steps * bs // n_learners for n in lrn)
    assert all(st[n]["updates"] == steps + 1 and st[n]["weight_group"]["size"] == n_engines + 1 for n in eng)  # (tp > 1: the size of ONE TP rank's group)
    assert sorted(r for n in eng for r in st[n]["weight_group"]["ranks"]) == list(range(1, 1 + n_engines * tp))
    assert s["engine_weights_equal_trainer_at_last_version"] is True
    assert s["weight_sync_under_load_ms"]["transport"] == "rccl_xgmi"
