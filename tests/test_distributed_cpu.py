"""World-size-2 tests on CPU (gloo): the N > 1 learner path (schedule -> streams -> LearnerStep
on two ranks with DDP) and the bucketed weight transfer protocol.  No GPU involved: the loss is an
injected torch function (the HIP loss needs a device) and the weight bytes travel over gloo."""

import json
import os
import socket
import types
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import preprocess as opre


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class TinyLM(torch.nn.Module):
    def __init__(self, vocab=32, dim=8):
        super().__init__()
        self.emb = torch.nn.Embedding(vocab, dim)
        self.head = torch.nn.Linear(dim, vocab)

    def forward(self, input_ids=None, **kw):
        return types.SimpleNamespace(logits=self.head(self.emb(input_ids)))


def _cpu_rl_step(model, batch, current_step, max_step, config, seq_parallel_group=None):
    """Stand-in loss with rl_step's signature: masked mean log-prob weighted by advantages."""
    logits = model(input_ids=batch.input_ids, attention_mask=batch.attention_mask, labels=batch.labels).logits
    lp = torch.log_softmax(logits[:, :-1].float(), -1).gather(2, batch.input_ids[:, 1:, None])[..., 0]
    mask = (batch.labels[:, 1:] != -100).float()
    loss = -(lp * batch.advantages[:, 1:] * mask).sum() / config.batch_size
    n = int(mask.sum().item())
    if n == 0:
        return loss, {"input_size": float(batch.input_ids.numel())}
    return loss, {"loss": loss.item(), "num_output_tokens_sum": n, "ratio_new_old_sum": float(n), "ratio_new_old_squared_sum": float(n)}


def _learner_rank(rank, world, port, exp_path, n_micro, out_dir):
    import torch.distributed as dist

    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune_loop import TRAINER_TOPIC, LearnerStep, batch_generator, run_data_loader
    import queue
    import threading

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    streams.set_streams_backend("files")
    torch.manual_seed(0)
    model = torch.nn.parallel.DistributedDataParallel(TinyLM())
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    step = LearnerStep(
        model, opt, RLConfig(policy_loss="ppo", kl_coef=0.0, final_kl_coef=0.0), train_batch_size=1,
        gradient_accumulation_passes=4, max_train_steps=100, send_weight_updates=False,
        trainer_stream=streams.SingleStreamSpec(exp_path=exp_path, topic=TRAINER_TOPIC), rl_step_fn=_cpu_rl_step,
    )
    assert step.samples_per_step == 4 and step.samples_per_lead_per_step == 2
    q = queue.Queue(maxsize=1)
    spec = streams.SingleStreamSpec(exp_path=exp_path, topic="training_data", partition=rank)
    threading.Thread(target=run_data_loader, args=(spec, q, None), daemon=True).start()
    gen = batch_generator(q)
    log = []
    for _ in range(n_micro):
        res = step.step(next(gen))
        log.append({"did": res["did_optimizer_step"], "samples": step.metrics.samples, "metrics": sorted(res["metrics"].keys())})
    step.finish()
    checksum = float(sum(p.detach().double().sum() for p in model.parameters()))
    Path(out_dir, f"rank{rank}.json").write_text(json.dumps({"log": log, "checksum": checksum, "steps": step.metrics.completed_steps,
                                                               "passes": step.metrics.passes}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_learner_follows_schedule(tmp_path):
    """Preprocessor side: schedule + collate + write per-rank streams.  Learner side: two gloo ranks
    read their partition, run LearnerStep.step() per micro-batch, step the optimizer on the same
    micro-batch index, end with identical (DDP-synchronised) parameters."""
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.finetune.utils import create_sentinel_batch
    from pipelinerl_amd.preprocess import MicroBatchScheduler
    from pipelinerl_amd.state import TrainerState
    from pipelinerl_amd.synthetic import make_entries

    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        raw = make_entries(3, attempts=4, seq_length=20, vocab=32, seed=4, prompt_min=2, prompt_max=5)
        data = opre.preprocess_chunk(raw, 2, False)
        sched = MicroBatchScheduler(num_trainers=2, train_batch_size=1, gradient_accumulation_passes=4, seq_length=40)
        sched.push(data)
        per_rank = {0: 0, 1: 0}
        out_spec = streams.StreamRangeSpec(exp_path=tmp_path, topic="training_data", partition_range=(0, 2))
        steps_done = 0
        with streams.write_to_streams(out_spec) as w:
            while sched.queue:
                mbs, done = sched.drain()
                for mb in mbs:
                    if mb.sentinel:
                        b = create_sentinel_batch(None, tokenizer=types.SimpleNamespace(eos_token_id=2), model_version=0)
                    else:
                        d = opre.collate_packed(mb.samples, 2, 1)
                        b = PipelineBatchEncoding(**{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()})
                    w.write(b, partition=mb.trainer_id)
                    per_rank[mb.trainer_id] += 1
                steps_done += int(done)
                if not mbs and not done:
                    break
        assert steps_done == 3 and per_rank[0] == per_rank[1]
        n_micro = per_rank[0]
        port = _free_port()
        mp.spawn(_learner_rank, args=(2, port, tmp_path, n_micro, tmp_path), nprocs=2, join=True)
        r0 = json.loads((tmp_path / "rank0.json").read_text())
        r1 = json.loads((tmp_path / "rank1.json").read_text())
        assert r0["steps"] == r1["steps"] == 3
        assert [e["did"] for e in r0["log"]] == [e["did"] for e in r1["log"]]
        assert r0["log"][-1]["samples"] == 12
        assert abs(r0["checksum"] - r1["checksum"]) < 1e-12  # DDP kept the replicas identical
        assert "rl/ess" in [k for e in r0["log"] for k in e["metrics"]]
        st = TrainerState(tmp_path)
        st.start_listening()
        assert st.wait_for_training_done(timeout=10) and st.samples_processed == 12
    finally:
        streams.reset_streams_backend()


class GlooBucketGroup:
    """Stands in for WeightSyncGroup on CPU: same `broadcast_bucket` contract over gloo."""

    def __init__(self, device):
        self.device = device

    def broadcast_bucket(self, bucket, mode="scatter_allgather", src=0):
        import torch.distributed as dist

        if mode == "scatter_allgather" and dist.get_world_size() > 2:
            raise NotImplementedError
        dist.broadcast(bucket, src=src)


def _wsync_rank(rank, world, port, out_dir):
    import torch.distributed as dist

    from pipelinerl_amd.weight_sync import BucketedReceiver, BucketedSender

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    src = TinyLM(vocab=300, dim=33)  # odd sizes: exercises the 256-byte alignment of the bucket plan
    src.head.weight.data = src.head.weight.data.to(torch.bfloat16).float()
    params = [(n, (p.detach().to(torch.bfloat16) if "head.weight" in n else p.detach())) for n, p in src.named_parameters()]
    grp = GlooBucketGroup(torch.device("cpu"))
    # the parameter list travels first (stands in for the HTTP WeightUpdateRequest), then the bytes
    objs = [[{"name": n, "shape": list(t.shape), "dtype": str(t.dtype)} for n, t in params] if rank == 0 else None]
    dist.broadcast_object_list(objs, src=0)
    if rank == 0:
        BucketedSender(grp, bucket_bytes=8192).send(params)
    if rank != 0:
        got = {}
        n = BucketedReceiver(grp, bucket_bytes=8192).receive(objs[0], lambda views: got.update({k: v.clone() for k, v in views}))
        ok = n == len(params) and all(torch.equal(got[name], t) and got[name].dtype == t.dtype for name, t in params)
        Path(out_dir, "wsync.json").write_text(json.dumps({"ok": bool(ok), "n": n}))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_weight_transfer_two_ranks(tmp_path):
    from pipelinerl_amd.weight_sync import ParamSpec, bucket_nbytes, plan_buckets

    specs = [ParamSpec("a", (1000,), torch.float32), ParamSpec("b", (3, 5), torch.bfloat16), ParamSpec("c", (5000,), torch.float32)]
    plan = plan_buckets(specs, bucket_bytes=8192)
    assert [[s.name for s, _ in b] for b in plan] == [["a", "b"], ["c"]]  # order kept, greedy fill, oversize gets its own bucket
    assert all(off % 256 == 0 for b in plan for _, off in b)
    assert bucket_nbytes(plan[0]) == 4096 + 256 and bucket_nbytes(plan[1]) == 20224
    port = _free_port()
    mp.spawn(_wsync_rank, args=(2, port, tmp_path), nprocs=2, join=True)
    res = json.loads((tmp_path / "wsync.json").read_text())
    assert res["ok"] and res["n"] == 3


class GlooGroup(GlooBucketGroup):
    """+ the per-tensor `broadcast` of the reference protocol."""

    def broadcast(self, tensor, src=0, stream=None):
        import torch.distributed as dist

        dist.broadcast(tensor, src=src)


def _weight_update_rank(rank, world, port, exp_path, transport):
    """rank 0: trainer with WeightUpdateManager; rank 1: an inference worker (WorkerExtension shim).
    The HTTP POST of the reference is replaced by a file drop that the worker polls."""
    import time

    import torch.distributed as dist

    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune_loop import TRAINER_TOPIC, WeightUpdateManager
    from pipelinerl_amd.vllm_worker import StandaloneWeightReceiver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    streams.set_streams_backend("files")
    request_file = Path(exp_path) / "request.json"
    torch.manual_seed(rank)  # different initial weights on the two sides
    model = TinyLM(vocab=50, dim=12)
    group = GlooGroup(torch.device("cpu"))
    if rank == 0:
        model.head.weight.data = model.head.weight.data.to(torch.bfloat16).float()
        named = lambda: [(n, p.to(torch.bfloat16) if n == "head.weight" else p) for n, p in model.named_parameters()]  # noqa: E731
        mgr = WeightUpdateManager(
            ["http://worker0"], model, streams.SingleStreamSpec(exp_path=exp_path, topic=TRAINER_TOPIC), group,
            is_main_process=True, named_parameters_fn=named, transport=transport, bucket_bytes=4096,
            post=lambda url, payload: request_file.write_text(json.dumps(payload)),
        )
        mgr.send_weight_update(version=32)
        mgr.shutdown()
        Path(exp_path, "trainer_checksum.json").write_text(json.dumps({n: float(p.double().sum()) for n, p in model.named_parameters()}))
    else:
        worker = StandaloneWeightReceiver(model, torch.device("cpu"), rank=0)
        worker.model_update_group = group
        worker.pg_rank = 1
        while not request_file.exists() or not request_file.read_text().endswith("}"):
            time.sleep(0.01)
        worker.receive_weight_update(request_file.read_text())
        Path(exp_path, "worker_checksum.json").write_text(json.dumps({n: float(p.double().sum()) for n, p in model.named_parameters()}))
        from pipelinerl_amd.finetune_loop import _barrier

        _barrier()  # matches the trainer-side barrier at the end of send_weight_update
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["bucketed", "per_tensor"])
def test_weight_update_protocol_two_ranks(tmp_path, transport):
    """WeightUpdateRequest -> bytes -> load_weights -> WeightUpdateSuccess, for the bucketed transport
    and for the reference's one-broadcast-per-parameter protocol."""
    import pytest as _pytest  # noqa: F401

    from pipelinerl_amd import streams
    from pipelinerl_amd.state import TrainerState

    port = _free_port()
    mp.spawn(_weight_update_rank, args=(2, port, tmp_path, transport), nprocs=2, join=True)
    a = json.loads((tmp_path / "trainer_checksum.json").read_text())
    b = json.loads((tmp_path / "worker_checksum.json").read_text())
    assert a.keys() == b.keys()
    for k in a:
        assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(a[k])), k
    req = json.loads((tmp_path / "request.json").read_text())
    assert req["kind"] == "weight_update_request" and req["version"] == 32 and req["transport"] == transport
    assert {p["name"]: p["dtype"] for p in req["parameters_info"]}["head.weight"] == "torch.bfloat16"
    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        st = TrainerState(tmp_path)
        st.start_listening()
        assert st.wait_for_model_version() == 32
    finally:
        streams.reset_streams_backend()


def test_unique_id_rendezvous_over_tcpstore(libprl):
    """Bootstrap of the weight-update group: rank 0 creates the RCCL unique id (no GPU needed for
    that) and publishes it through a TCPStore at `tcp://127.0.0.1:port`; the other ranks fetch the
    same 128 bytes (the communicator itself is created on GPUs only)."""
    import threading

    from pipelinerl_amd.weight_sync import WeightSyncGroup

    port = _free_port()
    out = {}

    def rank_fn(rank):
        out[rank] = WeightSyncGroup.exchange_unique_id(f"tcp://127.0.0.1:{port}", rank, 3, timeout_s=30)

    threads = [threading.Thread(target=rank_fn, args=(r,)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    uids = [out[r][0] for r in range(3)]
    assert len(uids[0]) == 128 and uids[0] == uids[1] == uids[2] and any(uids[0])


# ---------------------------------------------------------------------------------------------
# sharded trainers on the send side (reference finetune_loop.py:209-268)
# ---------------------------------------------------------------------------------------------


class _ShardedParam:
    """A ZeRO-3 style parameter: each trainer rank holds a flat slice, `ds_shape` is the full shape."""

    def __init__(self, full: torch.Tensor, rank: int, n_trainers: int):
        flat = full.reshape(-1)
        per = (flat.numel() + n_trainers - 1) // n_trainers
        self.ds_shape = tuple(full.shape)
        self.dtype = full.dtype
        self.shard = flat[rank * per:(rank + 1) * per].clone()
        self.per, self.numel = per, flat.numel()
        self.data = torch.empty(0, dtype=full.dtype)  # partitioned: no full tensor outside a gather

    @property
    def shape(self):
        return self.data.shape


def _zero3_rank(rank, world, port, out_dir, transport):
    """ranks 0, 1: trainers holding ZeRO-3 style shards; rank 2: the inference worker."""
    import contextlib

    import torch.distributed as dist

    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune_loop import WeightUpdateManager, Zero3Parameters
    from pipelinerl_amd.vllm_worker import StandaloneWeightReceiver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    trainers = dist.new_group([0, 1])
    update = dist.new_group([0, 2])  # rank 0 (trainer main) -> rank 2 (worker)
    streams.set_streams_backend("files")
    torch.manual_seed(0)
    full_model = TinyLM(vocab=300, dim=33)
    gathers = []

    class UpdateGroup(GlooBucketGroup):
        def broadcast_bucket(self, bucket, mode="scatter_allgather", src=0):
            dist.broadcast(bucket, src=0 if src == 0 else 2, group=update)

        def broadcast(self, tensor, src=0, stream=None):
            dist.broadcast(tensor, src=0, group=update)

    if rank < 2:
        sharded = {n: _ShardedParam(p.detach(), rank, 2) for n, p in full_model.named_parameters()}

        @contextlib.contextmanager
        def gathered(params):  # stands in for deepspeed.zero.GatheredParameters: a collective among the trainers
            gathers.append(len(params))
            for p in params:
                parts = [torch.zeros(p.per, dtype=p.dtype) for _ in range(2)]
                mine = torch.zeros(p.per, dtype=p.dtype)
                mine[: p.shard.numel()] = p.shard
                dist.all_gather(parts, mine, group=trainers)
                p.data = torch.cat(parts)[: p.numel].reshape(p.ds_shape)
            try:
                yield
            finally:
                for p in params:
                    p.data = torch.empty(0, dtype=p.dtype)

        engine = types.SimpleNamespace(module=types.SimpleNamespace(named_parameters=lambda: sharded.items()), zero_optimization_stage=lambda: 3)
        source = Zero3Parameters(engine, gathered=gathered)
        posted = []
        mgr = WeightUpdateManager(["http://worker"], engine, None, UpdateGroup(torch.device("cpu")), is_main_process=(rank == 0),
                                  transport=transport, bucket_bytes=1 << 16, post=lambda url, payload: posted.append(payload), parameter_source=source)
        import pipelinerl_amd.finetune_loop as fl

        fl._barrier = lambda: dist.barrier(group=trainers)  # the worker rank is not a trainer
        if rank == 0:
            # the HTTP request reaches the worker through the object channel of the update group
            desc = [{"name": n, "shape": list(shape), "dtype": str(dt)} for n, shape, dt in source.describe()]
            dist.broadcast_object_list([{"parameters_info": desc, "transport": transport, "bucket_bytes": 1 << 16, "version": 5}], src=0, group=update)
        mgr.send_weight_update(5)
        mgr.shutdown()
        Path(out_dir, f"trainer{rank}.json").write_text(json.dumps({"gathers": gathers, "posted": len(posted),
                                                                     "all_released": all(p.data.numel() == 0 for p in sharded.values())}))
    else:
        torch.manual_seed(1)
        worker_model = TinyLM(vocab=300, dim=33)  # different values: must end up equal to the trainers' model
        recv = StandaloneWeightReceiver(worker_model, torch.device("cpu"))
        recv.model_update_group = UpdateGroup(torch.device("cpu"))
        box = [None]
        dist.broadcast_object_list(box, src=0, group=update)
        recv.receive_weight_update({**box[0], "kind": "weight_update_request"})
        same = all(torch.equal(a, b) for (_, a), (_, b) in zip(worker_model.named_parameters(), full_model.named_parameters()))
        Path(out_dir, "worker.json").write_text(json.dumps({"same": bool(same)}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["bucketed", "per_tensor"])
def test_zero3_style_sharded_trainers_send_full_weights(tmp_path, transport):
    """Two trainer ranks hold flat shards of every parameter (ZeRO-3): the update gathers one BUCKET of
    parameters at a time on both ranks (collective), rank 0 flattens and sends, everything is released
    again; the worker ends up with the full weights."""
    port = _free_port()
    mp.spawn(_zero3_rank, args=(3, port, tmp_path, transport), nprocs=3, join=True)
    assert json.loads((tmp_path / "worker.json").read_text())["same"]
    t0 = json.loads((tmp_path / "trainer0.json").read_text())
    t1 = json.loads((tmp_path / "trainer1.json").read_text())
    assert t0["gathers"] == t1["gathers"] and t0["all_released"] and t1["all_released"]
    n_params = len(list(TinyLM(vocab=300, dim=33).named_parameters()))
    if transport == "bucketed":
        assert sum(t0["gathers"]) == n_params and len(t0["gathers"]) < n_params  # several parameters per gather
    else:
        assert t0["gathers"] == [1] * n_params  # the reference's schedule: one gather + one broadcast per parameter
    assert t0["posted"] == 1 and t1["posted"] == 0


def test_fsdp_source_drops_the_tied_head_and_dispatch_picks_the_source():
    from pipelinerl_amd.finetune_loop import FsdpParameters, PlainParameters, Zero3Parameters, parameter_source_for

    emb = torch.randn(5, 3)
    sd = {"model.embed_tokens.weight": emb, "model.norm.weight": torch.ones(3), "lm_head.weight": emb}
    src = FsdpParameters(object(), state_dict_fn=lambda: sd)
    assert [n for n, _, _ in src.describe()] == ["model.embed_tokens.weight", "model.norm.weight"]  # reference :258-262
    from pipelinerl_amd.weight_sync import ParamSpec

    with src.fetch([ParamSpec("model.norm.weight", (3,), torch.float32)]) as t:
        assert torch.equal(t["model.norm.weight"], torch.ones(3))
    src.release()
    assert isinstance(parameter_source_for(TinyLM()), PlainParameters)
    FullyShardedDataParallel = type("FullyShardedDataParallel", (), {})
    assert isinstance(parameter_source_for(FullyShardedDataParallel()), FsdpParameters)
    engine = types.SimpleNamespace(zero_optimization_stage=lambda: 3, module=TinyLM())
    try:
        import deepspeed  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            parameter_source_for(engine)
    else:
        assert isinstance(parameter_source_for(engine), Zero3Parameters)
    assert isinstance(parameter_source_for(types.SimpleNamespace(zero_optimization_stage=lambda: 2, named_parameters=lambda: [])), PlainParameters)


def _reduce_stats_rank(rank, world, port, out_dir):
    import torch.distributed as dist

    from pipelinerl_amd import _lib
    from pipelinerl_amd.hotpath import _MAX_LANES, _MIN_LANES, HotPathStep

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    step = HotPathStep.__new__(HotPathStep)  # the reduction needs no device state
    step.group = None
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(_lib.PRL_NUM_STATS, dtype=torch.float64, generator=g)
    got = step.reduce_stats(mine.clone())
    everyone = torch.stack([torch.randn(_lib.PRL_NUM_STATS, dtype=torch.float64, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)])
    want = everyone.sum(0)
    want[_MAX_LANES] = everyone[:, _MAX_LANES].max(0).values
    want[_MIN_LANES] = everyone[:, _MIN_LANES].min(0).values
    Path(out_dir, f"stats{rank}.json").write_text(json.dumps({"equal": bool(torch.equal(got, want)), "n": int(got.numel())}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_step_statistics_all_gather_at_world_8(tmp_path, world):
    """The ONE data-path collective of the sharded benchmark step (`HotPathStep.reduce_stats`, bench.py --gpus N): every rank ends with
    the same vector - sums of the additive lanes in rank order, max / min of the extrema lanes - at the 8 ranks the driver's scaling run uses."""
    port = _free_port()
    mp.spawn(_reduce_stats_rank, args=(world, port, tmp_path), nprocs=world, join=True)
    for r in range(world):
        d = json.loads((tmp_path / f"stats{r}.json").read_text())
        assert d["equal"] and d["n"] == 32, (r, d)


def _stateless_rank(rank, world, port, out_dir):
    from pipelinerl_amd.torch_utils import stateless_init_process_group

    grp = stateless_init_process_group(f"tcp://127.0.0.1:{port}", rank, world, "cpu", backend="gloo")
    assert grp.comm_size() == (world, rank)
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3).t() * (rank == 0)  # non-contiguous on purpose
    grp.broadcast(t, src=0)
    bucket = torch.full((1000,), rank, dtype=torch.uint8)
    grp.broadcast_bucket(bucket, mode="scatter_allgather")
    Path(out_dir, f"stateless{rank}.json").write_text(json.dumps({"t": t.tolist(), "bucket": int(bucket.sum()), "moved": grp.bytes_moved}))
    grp.close()


def test_stateless_weight_group_over_gloo_three_ranks(tmp_path):
    """`stateless_init_process_group(..., backend="gloo")` = `GlooWeightSyncGroup`: a group of its own (no default process group in these
    processes), the reference's per-tensor `.broadcast` and the bucket form, trainer rank 0 -> two workers."""
    port = _free_port()
    mp.spawn(_stateless_rank, args=(3, port, tmp_path), nprocs=3, join=True)
    want = (torch.arange(6, dtype=torch.float32).reshape(2, 3).t()).tolist()
    for r in range(3):
        d = json.loads((tmp_path / f"stateless{r}.json").read_text())
        assert d["t"] == want and d["bucket"] == 0 and d["moved"] == 24 + 1000
