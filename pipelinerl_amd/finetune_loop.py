"""Learner loop pieces: `LearnerStep.step()`, trainer messages, weight-update manager, data loader.

The reference has no `step()` function: the step is the body of the `while` loop of
`rl_finetuning_worker` (pipelinerl/finetune_loop.py:647-957).  `LearnerStep.step(batch)` has
exactly that contract:

    dequeue micro-batch -> count samples -> decide `do_optimizer_step` -> rl_step -> backward
    (sentinel: loss * 0) -> publish SamplesProcessed -> on boundary: optimizer step, lr step,
    metric aggregation; `maybe_send_weights()` applies the weight_update_interval rule (:936-949).

Differences, all inside the contract:
  * the per-micro-batch `dist.all_gather` of sample counters (:709, a latency-bound sync point)
    is a single int64 all-reduce;
  * `SamplesProcessed` goes through a persistent writer instead of open/append/close per
    micro-batch (:805-808);
  * gradient synchronisation is left to the wrapped model (DDP `no_sync`) exactly like
    `toggle_sync` (:746-755).
"""

from __future__ import annotations

import collections
import contextlib
import logging
import threading
import time
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor
from queue import Empty, Queue
from typing import Any, Callable, Literal

import numpy as np
import torch
from pydantic import BaseModel, Field, model_serializer

from .finetune.rl import RLConfig, rl_step as _rl_step
from .finetune.rl.utils import aggregate_rl_stats, effective_sample_size
from .finetune.types import PipelineBatchEncoding, TrainingMetrics
from .streams import SingleStreamSpec, read_stream, write_to_streams

logger = logging.getLogger(__name__)

TRAINER_TOPIC = "weight_update_request"  # carries every trainer message, not only weight updates


# ---------------------------------------------------------------------------------------------
# trainer -> everyone messages (reference finetune_loop.py:141-171; same `kind` discriminators)
# ---------------------------------------------------------------------------------------------


class ParameterInfo(BaseModel):
    name: str
    shape: list[int]
    dtype: str
    # transport "sharded" only (tp_shard.TpShard): the engine splits this parameter into `shard_parts` equal pieces along
    # `shard_dim`; None = replicated.  `shape` stays the FULL shape, as in the reference's message.
    shard_dim: int | None = None
    shard_parts: int = 1

    @model_serializer(mode="wrap")
    def _reference_wire_format(self, handler):
        """Uncut parameters serialise exactly as the reference's {name, shape, dtype} (finetune_loop.py:95-99): an
        unmodified reference receiver sees the message it knows."""
        d = handler(self)
        if self.shard_dim is None and self.shard_parts == 1:
            d.pop("shard_dim", None)
            d.pop("shard_parts", None)
        return d


class WeightUpdateRequest(BaseModel):
    kind: Literal["weight_update_request"] = "weight_update_request"
    version: int
    parameters_info: list[ParameterInfo]
    timestamp: float = Field(default_factory=time.time)
    # MI355X extensions: how the bytes travel ("per_tensor" = reference behaviour, "bucketed" = RCCL
    # buckets, "ipc" = trainer and worker share one GPU: the request carries HIP IPC handles,
    # "sharded" = RCCL buckets per tensor-parallel rank: a worker receives only its TP slices).
    # A request WITHOUT the field comes from an unmodified reference trainer, hence the default;
    # this package's WeightUpdateManager always states its transport.
    transport: str = "per_tensor"
    bucket_bytes: int = 1 << 30
    ipc_handles: list[str] = Field(default_factory=list)
    ipc_nbytes: list[int] = Field(default_factory=list)
    # the sender's IPC allocation cap: with bucket_bytes it fixes the row-range piece list (`weight_sync.split_for_ipc`), which
    # the receiver must derive identically - it travels with the request instead of being a constructor default on both sides
    ipc_max_allocation: int | None = None
    tp_size: int = 1


class WeightUpdateSuccess(BaseModel):
    kind: Literal["weight_update_success"] = "weight_update_success"
    version: int
    timestamp: float = Field(default_factory=time.time)


class SamplesProcessed(BaseModel):
    kind: Literal["samples_processed"] = "samples_processed"
    samples_processed: int
    timestamp: float = Field(default_factory=time.time)


class TrainingDone(BaseModel):
    kind: Literal["training_done"] = "training_done"
    timestamp: float = Field(default_factory=time.time)


TrainerMessage = WeightUpdateRequest | WeightUpdateSuccess | SamplesProcessed | TrainingDone

_MESSAGE_BY_KIND = {
    "weight_update_request": WeightUpdateRequest,
    "weight_update_success": WeightUpdateSuccess,
    "samples_processed": SamplesProcessed,
    "training_done": TrainingDone,
}


def parse_trainer_message(record: dict) -> TrainerMessage:
    try:
        return _MESSAGE_BY_KIND[record["kind"]](**record)
    except KeyError as e:
        raise ValueError(f"not a trainer message: {record!r}") from e


# ---------------------------------------------------------------------------------------------
# batch accounting helpers (reference :295-312)
# ---------------------------------------------------------------------------------------------


def gather_rl_metrics(rl_metrics: dict, group: Any = None) -> dict:
    """{metric: [values of this rank's micro-batches]} -> the same over every rank of `group`, finite values only (reference :64-89: one
    `all_gather_object` per optimizer step)."""
    import torch.distributed as dist

    parts: list = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, dict(rl_metrics), group=group)
    merged: dict[str, list] = defaultdict(list)
    for p in parts:
        for k, v in p.items():
            if v:
                merged[k].extend(x for x in v if np.isfinite(x))
    return merged


def validate_packing_config(args: Any) -> None:
    """Sequence packing needs an attention implementation that honours sequence boundaries (reference :315-323)."""
    if getattr(args, "seq_packing", False) and not getattr(args, "use_flash_attention", False):
        raise ValueError("Sequence packing requires flash attention. Either:\n"
                         "- Enable flash attention (use_flash_attention=true), or\n"
                         "- Disable sequence packing (seq_packing=false)")


def get_batch_token_count(batch: PipelineBatchEncoding) -> int:
    """Real (non-padding) tokens of a batch."""
    return int(batch.attention_mask.sum().item())


def get_batch_sequence_count(batch: PipelineBatchEncoding) -> int:
    """Sequences in a batch; the sequence-parallel filler of a packed batch does not count."""
    if batch.position_ids is not None:
        assert batch.seq_boundaries is not None
        return len(batch.seq_boundaries) - (1 if batch.padding == 0 else 2)
    return batch.input_ids.size(0)


def calculate_train_steps(args: Any, interrupt_train_steps: int) -> int:
    """reference :1100-1107."""
    if interrupt_train_steps == -1:
        assert args.interrupt_train_steps <= args.max_train_steps
        return args.max_train_steps if args.interrupt_train_steps < 0 else args.interrupt_train_steps
    assert interrupt_train_steps <= args.max_train_steps
    return interrupt_train_steps


# ---------------------------------------------------------------------------------------------
# data loader thread (reference :92-134)
# ---------------------------------------------------------------------------------------------


def annotate_host_batch(batch: PipelineBatchEncoding) -> PipelineBatchEncoding:
    """What the loader thread knows about a batch while it is still on the HOST, stored in `batch.model_extra` so that the
    training thread never has to ask the device for it: `tokens` (the real-token count, `get_batch_token_count`) and
    `labelled_rows` (flat indices of the logits rows that predict a labelled token, `fused_head._labelled_rows`)."""
    if batch.input_ids.is_cuda:
        return batch
    batch.model_extra["tokens"] = int(batch.attention_mask.sum())
    live = torch.zeros_like(batch.labels, dtype=torch.bool)
    live[:, :-1] = batch.labels[:, 1:] != -100
    batch.model_extra["labelled_rows"] = live.flatten().nonzero().squeeze(1)
    return batch


class RecordToBatch:
    """One `training_data` record -> the batch on `device`, as the loader thread does it: a full-wire record (the kwargs of a
    `PipelineBatchEncoding`) is rebuilt and its columns copied over; a compact-wire record (`finetune.data.CompactBatch`: the
    micro-batch before expansion, `PreprocessorLoop(wire="compact")`) is EXPANDED here - one upload through a page-locked ring
    + one pack-kernel launch on `device`.  `annotate`: leave the host-side facts `StreamedLearnerStep` wants in
    `batch.model_extra` (`annotate_host_batch`)."""

    def __init__(self, device: Any, annotate: bool = False):
        self.device, self.annotate = device, annotate
        self._stager = None

    def __call__(self, record: Any) -> PipelineBatchEncoding:
        from .finetune.data import CompactBatch

        device = self.device
        if isinstance(record, CompactBatch):
            if device is None or torch.device(device).type != "cuda":
                raise RuntimeError("a compact training_data record is expanded by the pack kernel: the loader needs the learner's HIP device")
            if self._stager is None:
                from .staging import PinnedStager

                self._stager = PinnedStager(device, slots=4)
            with torch.cuda.device(device):
                batch = record.to_batch(device, self._stager)
            if self.annotate:
                facts = record.host_facts()
                batch.model_extra["tokens"] = facts["tokens"]
                batch.model_extra["labelled_rows"] = facts["labelled_rows"].to(device, non_blocking=True)
            return batch
        batch = PipelineBatchEncoding(**record)
        if self.annotate:
            annotate_host_batch(batch)
        if device is not None:
            batch = batch.to_device(device)
            rows = batch.model_extra.get("labelled_rows")
            if rows is not None:
                batch.model_extra["labelled_rows"] = rows.to(device, non_blocking=True)
        return batch


def run_data_loader(data_stream: SingleStreamSpec, batch_queue: Queue, device: Any, stop: threading.Event | None = None,
                    annotate: bool = False) -> None:
    """Read `training_data/<instance>/<rank>` records, rebuild the batch, move it to `device`,
    hand it to the training thread.  Exceptions travel through the queue like in the reference.
    Both wires are understood (`RecordToBatch`)."""
    try:
        to_batch = RecordToBatch(device, annotate)
        with read_stream(data_stream) as reader:
            for record in reader.read():
                if stop is not None and stop.is_set():
                    return
                batch_queue.put(to_batch(record))
    except Exception as e:  # noqa: BLE001 - forwarded to the consumer
        logger.error(f"Error in stream reader: {e}")
        batch_queue.put(e)


def batch_generator(batch_queue: Queue, stop: threading.Event | None = None):
    """Blocking iterator over the loader queue with the reference's growing poll timeout (:583-597)."""
    while True:
        timeout = 0.1
        while True:
            try:
                item = batch_queue.get(timeout=timeout)
                break
            except Empty:
                if stop is not None and stop.is_set():
                    return
                timeout = min(timeout * 1.5, 5.0)
        if isinstance(item, Exception):
            raise item
        yield item


# ---------------------------------------------------------------------------------------------
# where the trainer's FULL parameters come from (reference :205-268)
# ---------------------------------------------------------------------------------------------


class ParameterSource:
    """Full (unsharded) parameters of the trained model for a weight update.

    `describe()` - names, FULL shapes and dtypes, computable on every rank without communication;
    `fetch(specs)` - context manager giving {name: full tensor} for the listed parameters on the main rank.
    For sharded trainers entering it is a COLLECTIVE: every trainer rank calls `fetch` with the same
    lists in the same order (`WeightUpdateManager.send_weight_update` does that)."""

    def describe(self) -> list[tuple[str, tuple[int, ...], torch.dtype]]:
        raise NotImplementedError

    def fetch(self, specs: list) -> Any:
        raise NotImplementedError


class PlainParameters(ParameterSource):
    """DDP / single GPU: the unwrapped module's `named_parameters()` (reference :263-265)."""

    def __init__(self, model: Any, named_parameters_fn: Callable | None = None):
        self._fn = named_parameters_fn or (lambda: _unwrap(model).named_parameters())

    def _params(self) -> dict[str, torch.Tensor]:
        return {n: p.detach() for n, p in self._fn()}

    def describe(self):
        return [(n, tuple(p.shape), p.dtype) for n, p in self._params().items()]

    @contextlib.contextmanager
    def fetch(self, specs):
        params = self._params()
        yield {sp.name: params[sp.name] for sp in specs}


class Zero3Parameters(ParameterSource):
    """DeepSpeed ZeRO stage 3: every parameter is partitioned over the trainer ranks; its full shape is
    `parameter.ds_shape` and `deepspeed.zero.GatheredParameters` materialises it (reference :209-238, one
    parameter at a time).  Here a whole BUCKET of parameters is gathered at once: 15 collectives for
    Qwen2.5-7B instead of 339.  `gathered` is injectable so the protocol can be exercised without
    DeepSpeed (tests/test_distributed_cpu.py)."""

    def __init__(self, engine: Any, gathered: Callable | None = None):
        module = engine.module if hasattr(engine, "module") else engine
        module = getattr(module, "pretrained_model", module)  # value-head wrapper: only the policy travels (:217-221)
        self._named = dict(module.named_parameters())
        if gathered is None:
            import deepspeed  # noqa: PLC0415 - optional dependency

            gathered = deepspeed.zero.GatheredParameters
        self._gathered = gathered

    def describe(self):
        return [(n, tuple(getattr(p, "ds_shape", p.shape)), p.dtype) for n, p in self._named.items()]

    @contextlib.contextmanager
    def fetch(self, specs):
        params = [self._named[sp.name] for sp in specs]
        with self._gathered(params):
            yield {sp.name: p.data for sp, p in zip(specs, params)}


class FsdpParameters(ParameterSource):
    """torch FSDP: the FULL_STATE_DICT gathered to rank 0 (reference :250-262).  A tied `lm_head.weight`
    shows up in the state dict although it is not a parameter of its own; it is dropped like the
    reference does (:258-262) so that the receiver does not load the embedding twice."""

    def __init__(self, model: Any, state_dict_fn: Callable | None = None):
        self.model = model
        self._state_dict_fn = state_dict_fn
        self._cache: dict[str, torch.Tensor] | None = None

    def _full_state(self) -> dict[str, torch.Tensor]:
        if self._state_dict_fn is not None:
            sd = dict(self._state_dict_fn())
        else:
            from torch.distributed.fsdp import FullStateDictConfig, FullyShardedDataParallel as FSDP, StateDictType

            with FSDP.state_dict_type(self.model, StateDictType.FULL_STATE_DICT, FullStateDictConfig(offload_to_cpu=False, rank0_only=True)):
                sd = dict(self.model.state_dict())
        if "lm_head.weight" in sd:
            logger.info("Removing lm_head.weight from gathered parameters, because it's not a parameter.")
            del sd["lm_head.weight"]
        return sd

    def gather(self) -> None:
        """Collective: every trainer rank calls it once per update; rank 0 keeps the result until `release`."""
        self._cache = self._full_state()

    def release(self) -> None:
        self._cache = None

    def describe(self):
        if self._cache is None:
            self.gather()
        return [(n, tuple(t.shape), t.dtype) for n, t in self._cache.items()]

    @contextlib.contextmanager
    def fetch(self, specs):
        yield {sp.name: self._cache[sp.name] for sp in specs}


def parameter_source_for(model: Any, named_parameters_fn: Callable | None = None) -> ParameterSource:
    """The reference's dispatch (:209-212, :250): DeepSpeed engine at ZeRO stage 3, FSDP, or plain."""
    if named_parameters_fn is not None:
        return PlainParameters(model, named_parameters_fn)
    stage = getattr(model, "zero_optimization_stage", None)
    if callable(stage) and stage() == 3:
        return Zero3Parameters(model)
    if type(model).__name__ == "FullyShardedDataParallel":
        return FsdpParameters(model)
    return PlainParameters(model)


# ---------------------------------------------------------------------------------------------
# weight updates, send side (reference :174-292)
# ---------------------------------------------------------------------------------------------


class WeightUpdateManager:
    """Trainer side of the in-flight weight update: rank 0 announces the parameter list to every
    inference server over HTTP (the POST returns once the update is applied, vllm1.py:244-249),
    streams the bytes through the RCCL update group, then publishes `WeightUpdateSuccess`.

    `actor_update_group` is a `pipelinerl_amd.weight_sync.WeightSyncGroup`; `named_parameters_fn`
    returns the full (gathered) parameters on rank 0 — e.g. `lambda: model.named_parameters()`
    for DDP, or a FULL_STATE_DICT gather for FSDP."""

    def __init__(self, llm_urls: list[str], accelerated_model: Any, update_stream: SingleStreamSpec | None,
                 actor_update_group: Any, is_main_process: bool = True, named_parameters_fn: Callable | None = None,
                 transport: str = "bucketed", bucket_bytes: int = 1 << 30, post: Callable | None = None,
                 parameter_source: ParameterSource | None = None, tp_shards: dict | None = None, kv_heads: int | None = None):
        """`transport="sharded"`: `actor_update_group` is the LIST of per-TP-rank communicators
        (`WeightSyncGroup.tp_shard_groups`); every worker receives only the slices its TP rank stores.  The cut of each
        parameter comes from `tp_shard.plan_tp_shards` (Llama / Qwen layout; `kv_heads` for grouped-query models with
        fewer KV heads than TP ranks) or from `tp_shards` (name -> TpShard) for other engines."""
        self.llm_urls = llm_urls
        self.accelerated_model = accelerated_model
        self.update_stream = update_stream
        self.actor_update_group = actor_update_group
        self.is_main_process = is_main_process
        self.source = parameter_source or parameter_source_for(accelerated_model, named_parameters_fn)
        self.transport = transport
        self.bucket_bytes = bucket_bytes
        self.tp_shards, self.kv_heads = tp_shards, kv_heads
        self.thread_pool = ThreadPoolExecutor(max_workers=max(1, len(llm_urls)))
        self._post = post or _http_post
        self._sender = None
        self._shutdown = False

    def _request_weight_update(self, url: str, message: WeightUpdateRequest) -> None:
        try:
            self._post(url + "/receive_weight_update", message.model_dump())
        except Exception as e:  # noqa: BLE001 - logged, not raised (reference :188-192)
            logger.error(f"Error sending weight update request to {url}: {e}")

    def request_weight_updates(self, message: WeightUpdateRequest):
        return [self.thread_pool.submit(self._request_weight_update, url, message) for url in self.llm_urls]

    def shutdown(self) -> None:
        if not self._shutdown:
            self.thread_pool.shutdown(wait=True)
            self._shutdown = True

    def send_weight_update(self, version: int) -> None:
        """Blocking; every trainer rank calls it (gathering sharded parameters is a collective among
        them), rank 0 sends."""
        from .weight_sync import BucketedSender, ParamSpec, plan_buckets

        src = self.source
        if isinstance(src, FsdpParameters):
            src.gather()  # collective: FULL_STATE_DICT to rank 0
        described = src.describe()
        specs = [ParamSpec(n, tuple(shape), dt) for n, shape, dt in described]
        futures = []
        shards, tp_size = None, 1
        if self.transport == "sharded":
            from .tp_shard import plan_tp_shards

            groups = self.actor_update_group if isinstance(self.actor_update_group, (list, tuple)) else [self.actor_update_group]
            tp_size = len(groups)
            shards = plan_tp_shards([(sp.name, sp.shape) for sp in specs], tp_size, self.kv_heads, self.tp_shards)
        if self.is_main_process:
            info = [ParameterInfo(name=sp.name, shape=list(sp.shape), dtype=str(sp.dtype),
                                  shard_dim=shards[sp.name].dim if shards else None, shard_parts=shards[sp.name].parts if shards else 1)
                    for sp in specs]
            message = WeightUpdateRequest(version=version, parameters_info=info, transport=self.transport, bucket_bytes=self.bucket_bytes,
                                          tp_size=tp_size)
            if self.transport == "ipc":
                # colocated: fill the exported buckets first, the request then carries their handles
                from .weight_sync import ColocatedSender

                with src.fetch(specs) as tensors:
                    params = [(sp.name, tensors[sp.name]) for sp in specs]
                    if self._sender is None:
                        self._sender = ColocatedSender(params[0][1].device, self.bucket_bytes)
                    desc = self._sender.publish(params)
                message.ipc_handles, message.ipc_nbytes = desc["ipc_handles"], desc["ipc_nbytes"]
                message.ipc_max_allocation = desc["ipc_max_allocation"]
            futures = self.request_weight_updates(message)
        if self.transport == "ipc":
            # the POST returns when the worker has copied the buckets.  The other trainer ranks of a SHARDED
            # source still have to enter the gather the main rank ran above: `fetch` is a collective there
            # (GatheredParameters / the FSDP state dict), a rank that skips it deadlocks the step.
            if not self.is_main_process and not isinstance(src, PlainParameters):
                with src.fetch(specs):
                    pass
        elif self.transport in ("bucketed", "sharded"):
            # a sharded update gathers in groups of bucket_bytes * tp_size of FULL parameters: one bucket per TP rank each
            gather_bytes = self.bucket_bytes * tp_size
            if self.is_main_process:
                if self._sender is None:
                    from .weight_sync import ShardedSender

                    self._sender = (ShardedSender(groups, self.bucket_bytes) if self.transport == "sharded"
                                    else BucketedSender(self.actor_update_group, self.bucket_bytes))
                if self.transport == "sharded":
                    self._sender.send_streamed(specs, shards, src.fetch)
                else:
                    self._sender.send_streamed(specs, src.fetch)
            elif not isinstance(src, PlainParameters):
                # the other trainer ranks take part in the same gathers, bucket by bucket
                for bucket in plan_buckets(specs, gather_bytes):
                    with src.fetch([sp for sp, _ in bucket]):
                        pass
        else:  # the reference's one-broadcast-per-parameter protocol (:230-238, :276-282)
            for sp in specs:
                if self.is_main_process or not isinstance(src, PlainParameters):
                    with src.fetch([sp]) as tensors:
                        if self.is_main_process:
                            self.actor_update_group.broadcast(tensors[sp.name].data, src=0)
        if isinstance(src, FsdpParameters):
            src.release()
        if self.is_main_process:
            for f in futures:
                f.result()
            if self.update_stream is not None:
                with write_to_streams(self.update_stream) as writer:
                    writer.write(WeightUpdateSuccess(version=version))
        _barrier()


def _http_post(url: str, payload: dict) -> None:
    import requests

    r = requests.post(url, json=payload)
    r.raise_for_status()


def _unwrap(model: Any) -> Any:
    return getattr(model, "module", model)


def _barrier() -> None:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


# ---------------------------------------------------------------------------------------------
# LearnerStep
# ---------------------------------------------------------------------------------------------


class LearnerStep:
    """One data-parallel learner rank.  `step(batch)` consumes one micro-batch."""

    def __init__(
        self,
        model: torch.nn.Module,
        optimizer: torch.optim.Optimizer,
        rl_config: RLConfig,
        train_batch_size: int,
        gradient_accumulation_passes: int,
        max_train_steps: int,
        lr_scheduler: Any = None,
        seq_parallel: int = 1,
        weight_update_manager: WeightUpdateManager | None = None,
        weight_update_interval: int = 1,
        send_weight_updates: bool = True,
        trainer_stream: SingleStreamSpec | None = None,
        training_metrics: TrainingMetrics | None = None,
        gradient_clipping_threshold: float | None = None,
        max_lag: int | None = None,
        rl_step_fn: Callable = _rl_step,
        process_group: Any = None,
        seq_parallel_group: Any = None,
    ):
        """`seq_parallel_group`: the ranks that hold the slices of one packed sequence (`seq_parallel` > 1);
        forwarded to `rl_step` like the reference does (finetune_loop.py:768-775) - GSPO needs it to add the
        per-segment sums over the slices."""
        import torch.distributed as dist

        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.rl_step_fn = rl_step_fn
        self.group = process_group
        self.seq_parallel_group = seq_parallel_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.rank = dist.get_rank(process_group) if self.distributed else 0
        self.seq_parallel = seq_parallel
        self.metrics = training_metrics or TrainingMetrics()
        self.max_train_steps = max_train_steps
        self.weight_update_manager = weight_update_manager
        self.weight_update_interval = weight_update_interval
        self.send_weight_updates = send_weight_updates
        self.gradient_clipping_threshold = gradient_clipping_threshold
        self.max_lag = max_lag
        self.train_batch_size = train_batch_size

        # sample accounting (reference :627-646)
        num_lead = self.world // seq_parallel
        passes_per_lead = gradient_accumulation_passes // num_lead
        self.samples_per_lead_per_step = passes_per_lead * train_batch_size
        self.samples_per_step = self.samples_per_lead_per_step * num_lead
        self.start_samples = self.metrics.samples
        self.target_samples_per_lead = self.samples_per_lead_per_step
        self.target_samples = self.samples_per_step
        self.local_samples = 0
        self.total_samples = 0
        self.rl_config = rl_config.model_copy()
        self.rl_config.batch_size = self.samples_per_step  # the loss normaliser (:644-646)

        self._rl_metrics: dict[str, list] = defaultdict(list)
        self._micro_batch_sizes: list[int] = []
        self._tokens: list[int] = []
        self._lag: dict[str, int] = {}
        self._writer_cm = None
        self._writer = None
        if trainer_stream is not None and self.rank == 0:
            self._writer_cm = write_to_streams(trainer_stream)
            self._writer = self._writer_cm.__enter__()
        self._counter_device = next(model.parameters()).device if any(True for _ in model.parameters()) else torch.device("cpu")
        # The per-micro-batch sample counter (finetune_loop.py:709) is summed over the ranks on the HOST.  Under RCCL a device tensor + `.item()`
        # would drain the GPU's queue once per micro-batch - the synchronisation `StreamedLearnerStep` exists to avoid - so the default
        # group gets a gloo companion for these 8 bytes (every rank constructs its step, so every rank takes part in creating it); a
        # caller-supplied subgroup keeps its own backend.
        self._counter_group = process_group
        if self.distributed and self.world > 1 and process_group is None and dist.get_backend() == "nccl":
            self._counter_group = dist.new_group(backend="gloo")

    # -- helpers --------------------------------------------------------------------------------
    def _sum_over_ranks(self, value: int) -> int:
        if not self.distributed or self.world == 1:
            return value
        import torch.distributed as dist

        # (a gloo group reduces host memory: a device tensor would be staged through the host behind a stream synchronisation,
        # i.e. drain the GPU once per micro-batch; only RCCL needs the counter on the device)
        device = "cpu" if dist.get_backend(self._counter_group) == "gloo" else self._counter_device
        t = torch.tensor([value], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._counter_group)
        return int(t.item())

    @contextlib.contextmanager
    def _sync_context(self, sync: bool):
        if sync or not hasattr(self.model, "no_sync"):
            yield
        else:
            with self.model.no_sync():
                yield

    def publish(self, message: BaseModel) -> None:
        if self._writer is not None:
            self._writer.write(message)

    # -- the step -------------------------------------------------------------------------------
    def step(self, batch: PipelineBatchEncoding) -> dict[str, Any]:
        """Consume one micro-batch.  Returns {"loss", "did_optimizer_step", "stats", "metrics"}."""
        m = self.metrics
        is_sentinel = bool(batch.sentinel)
        if self.local_samples == self.target_samples_per_lead:
            assert is_sentinel, "We should get a sentinel batch"
        if self.max_lag is not None and m.last_broadcasted_version - batch.model_version > self.max_lag:
            m.samples_too_old_to_train += self.train_batch_size
        self._lag["min_version"] = min(self._lag.get("min_version", batch.model_version), batch.model_version)
        self._lag["max_version"] = max(self._lag.get("max_version", batch.model_version), batch.model_version)

        if not is_sentinel:
            m.passes += 1
            n = get_batch_sequence_count(batch)
            self._micro_batch_sizes.append(n)
            self.local_samples += n
            self._tokens.append(get_batch_token_count(batch))

        overcounted = self._sum_over_ranks(self.local_samples)
        assert overcounted % self.seq_parallel == 0
        self.total_samples = overcounted // self.seq_parallel
        do_optimizer_step = self.total_samples == self.target_samples

        with self._sync_context(do_optimizer_step):
            if self.seq_parallel_group is not None:
                loss, stats = self.rl_step_fn(self.model, batch, m.completed_steps, self.max_train_steps, self.rl_config,
                                              seq_parallel_group=self.seq_parallel_group)
            else:
                loss, stats = self.rl_step_fn(self.model, batch, m.completed_steps, self.max_train_steps, self.rl_config)
            if is_sentinel:
                loss = loss * 0.0  # keeps every rank's forward/backward count equal, adds nothing
            else:
                for k, v in stats.items():
                    self._rl_metrics[k].append(v)
            loss.backward()

        self.publish(SamplesProcessed(samples_processed=self.start_samples + self.total_samples))
        result: dict[str, Any] = {"loss": loss.detach(), "did_optimizer_step": False, "stats": stats, "metrics": {}}
        if not do_optimizer_step:
            return result

        # ---- accumulation boundary: optimizer step (:810-857)
        self.target_samples_per_lead += self.samples_per_lead_per_step
        self.target_samples += self.samples_per_step
        m.completed_steps += 1
        m.samples = self.start_samples + self.total_samples
        m.tokens += sum(self._tokens) * self.world
        assert sum(self._micro_batch_sizes) == self.samples_per_lead_per_step
        if self.gradient_clipping_threshold is not None:
            gn = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.gradient_clipping_threshold)
            m.grad_norm = float(gn)
        self.optimizer.step()
        self.optimizer.zero_grad()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        result["did_optimizer_step"] = True
        result["metrics"] = self._aggregate_metrics()
        self._rl_metrics = defaultdict(list)
        self._micro_batch_sizes, self._tokens, self._lag = [], [], {}
        # the local counter keeps running across steps, like the reference's `local_samples`
        return result

    def _aggregate_metrics(self) -> dict[str, float]:
        """rl/* metrics of the finished step over all ranks (reference :908-922)."""
        gathered = dict(self._rl_metrics)
        if self.distributed and self.world > 1:
            gathered = gather_rl_metrics(gathered, self.group)
        if not gathered:
            return {}
        avg = aggregate_rl_stats(gathered, self.samples_per_step)
        if all(k in avg for k in ("rl/ratio_new_old_sum", "rl/ratio_new_old_squared_sum", "rl/num_output_tokens_sum")):
            avg["rl/ess"] = effective_sample_size(avg)
        return avg

    def maybe_send_weights(self) -> bool:
        """After an optimizer step: broadcast when enough samples were trained since the last
        broadcast (`weight_update_interval`, reference :936-949).  Model version == samples trained."""
        m = self.metrics
        if not self.send_weight_updates or self.weight_update_manager is None:
            return False
        if m.samples - m.last_broadcasted_version < self.weight_update_interval:
            return False
        self.weight_update_manager.send_weight_update(m.samples)
        m.last_broadcasted_version = m.samples
        return True

    def finish(self) -> None:
        """Signal `TrainingDone` and close the trainer stream."""
        self.publish(TrainingDone())
        if self._writer_cm is not None:
            self._writer_cm.__exit__(None, None, None)
            self._writer_cm = self._writer = None


class StreamedLearnerStep(LearnerStep):
    """`LearnerStep` for a model prepared with `fused_head.install_fused_head`, fed from the `training_data` stream, WITHOUT a
    host synchronisation per micro-batch: the same contract (sample accounting, sentinel rule, `SamplesProcessed`, optimizer
    step at the accumulation boundary, `maybe_send_weights`), but

      * the loss comes from the model's own forward (`model(rl_batch=...)` -> loss, device statistics): hidden states ->
        MFMA head -> K2+K3, no `[T, V]` logits;
      * the statistics of a micro-batch STAY ON THE DEVICE until the accumulation boundary, where all of them come over in
        one copy and go through the reference's asserts and its per-step aggregation (`check_finite`, `aggregate_rl_stats`);
        `step()` of a micro-batch inside the step returns `stats = None`.  The reference pays ~31 `.item()` per micro-batch
        (rl/__init__.py:398-439), `LearnerStep` one copy per micro-batch; with a 0.5B model at 2048 tokens per micro-batch the
        GPU work of one is short enough that ANY per-micro-batch sync leaves the device idle while the host queues the next;
      * token counts and the rows that carry a label come from the loader thread, which sees the batch on the HOST before it
        uploads it (`batch.model_extra["tokens" / "labelled_rows"]`, `run_data_loader(..., annotate=True)`) - the two
        remaining per-micro-batch syncs of the drop-in path (`attention_mask.sum().item()`, `nonzero()`).

    A non-finite value is therefore reported at the END of the step it occurred in, not at its micro-batch."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        inner = self.model
        while getattr(inner, "_prl_fused_head", None) is None and hasattr(inner, "module"):
            inner = inner.module
        if getattr(inner, "_prl_fused_head", None) is None:
            raise TypeError("StreamedLearnerStep drives a model prepared with pipelinerl_amd.fused_head.install_fused_head")
        if self.seq_parallel != 1 and self.seq_parallel_group is None:
            # slices of a packed sequence (types.py:145-180) trained without their group would silently drop the sequence-parallel
            # reduction of the sequence-level sums (rl/utils.py:194-206); the group is forwarded to the model's forward in `step`
            raise ValueError("StreamedLearnerStep with seq_parallel > 1 needs seq_parallel_group (the ranks that hold the slices of one sequence)")
        self._stats_dev: list[torch.Tensor] = []
        self._input_sizes: list[int] = []
        # per real micro-batch: samples trained so far - the batch's model version; the most recent 4096 (a run is unbounded)
        self.lag_samples: collections.deque[int] = collections.deque(maxlen=4096)

    def step(self, batch: PipelineBatchEncoding) -> dict[str, Any]:
        from .finetune.rl import VALUE_STAT_KEYS, check_finite, make_loss_config, stats_to_dict
        from . import _lib

        m = self.metrics
        is_sentinel = bool(batch.sentinel)
        if self.local_samples == self.target_samples_per_lead:
            assert is_sentinel, "We should get a sentinel batch"
        if self.max_lag is not None and m.last_broadcasted_version - batch.model_version > self.max_lag:
            m.samples_too_old_to_train += self.train_batch_size
        self._lag["min_version"] = min(self._lag.get("min_version", batch.model_version), batch.model_version)
        self._lag["max_version"] = max(self._lag.get("max_version", batch.model_version), batch.model_version)
        if not is_sentinel:
            m.passes += 1
            n = get_batch_sequence_count(batch)
            self._micro_batch_sizes.append(n)
            self.local_samples += n
            tokens = batch.model_extra.get("tokens") if hasattr(batch, "model_extra") else None
            self._tokens.append(int(tokens) if tokens is not None else get_batch_token_count(batch))
            self.lag_samples.append(m.samples - int(batch.model_version))
        overcounted = self._sum_over_ranks(self.local_samples)
        assert overcounted % self.seq_parallel == 0
        self.total_samples = overcounted // self.seq_parallel
        do_optimizer_step = self.total_samples == self.target_samples

        with self._sync_context(do_optimizer_step):
            sp = {"seq_parallel_group": self.seq_parallel_group} if self.seq_parallel_group is not None else {}  # (finetune_loop.py:768-775)
            loss, stats_dev = self.model(rl_batch=batch, rl_config=self.rl_config, current_step=m.completed_steps, max_step=self.max_train_steps, **sp)
            if is_sentinel:
                loss = loss * 0.0
            else:
                self._stats_dev.append(stats_dev)
                self._input_sizes.append(int(batch.input_ids.numel()))
            loss.backward()
        self.publish(SamplesProcessed(samples_processed=self.start_samples + self.total_samples))
        result: dict[str, Any] = {"loss": loss.detach(), "did_optimizer_step": False, "stats": None, "metrics": {}}
        if not do_optimizer_step:
            return result

        # ---- accumulation boundary: the step's statistics in ONE copy, then the optimizer step (:810-857)
        _, kl_coef, ent_coef = make_loss_config(self.rl_config, m.completed_steps, self.max_train_steps)
        if self._stats_dev:
            rows = torch.stack(self._stats_dev).cpu().tolist()
            for row, size in zip(rows, self._input_sizes):
                vstats = row[_lib.PRL_NUM_STATS:]
                if vstats:
                    row[_lib.STAT_INDEX["loss"]] = float(np.float32(row[_lib.STAT_INDEX["loss"]]) + np.float32(
                        self.rl_config.value_loss_coef * np.float32(vstats[VALUE_STAT_KEYS.index("value_loss")])))
                check_finite(row)
                if int(row[_lib.STAT_INDEX["num_output_tokens_sum"]]) == 0:
                    stats = {"input_size": float(size)}
                else:
                    stats = stats_to_dict(row, kl_coef, ent_coef, size)
                    stats.update({k: float(np.float32(v)) for k, v in zip(VALUE_STAT_KEYS, vstats)})
                for k, v in stats.items():
                    self._rl_metrics[k].append(v)
            result["stats"] = stats
        self._stats_dev, self._input_sizes = [], []
        self.target_samples_per_lead += self.samples_per_lead_per_step
        self.target_samples += self.samples_per_step
        m.completed_steps += 1
        m.samples = self.start_samples + self.total_samples
        m.tokens += sum(self._tokens) * self.world
        assert sum(self._micro_batch_sizes) == self.samples_per_lead_per_step
        if self.gradient_clipping_threshold is not None:
            gn = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.gradient_clipping_threshold)
            m.grad_norm = float(gn)
        self.optimizer.step()
        self.optimizer.zero_grad()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        result["did_optimizer_step"] = True
        result["metrics"] = self._aggregate_metrics()
        self._rl_metrics = defaultdict(list)
        self._micro_batch_sizes, self._tokens, self._lag = [], [], {}
        return result


class NativeLearnerStep:
    """The MI355X-native optimizer step: consumes a whole step's rollouts at once.

        K5 + ONE K6 launch  ->  per micro-batch: model forward, fused logits kernel (loss gradient
        straight into d logits), model backward  ->  ONE K2+K3 launch for loss + 32 statistics
        ->  one stats all-gather  ->  optimizer step

    No host synchronisation happens inside the step; `step()` returns device tensors and only
    `stats_dict()` copies 256 bytes to the host.  Numerically identical to running the drop-in
    `rl_step` per micro-batch and summing (tests/test_gpu_pipeline.py).

    The `step()` contract of the reference loop (finetune_loop.py:647-957) is kept at STEP granularity:

      * sample accounting (:627-646, :698-713): every rank passes its share of the step; ONE all-gather
        of (micro-batches, samples) per step checks that the shares add up to `samples_per_step` and
        tells every rank the largest micro-batch count - ranks with fewer run sentinel micro-batches
        (zero loss, full forward/backward) so that sharded trainers, whose every forward/backward is
        a collective, stay in lock-step (:599-607, :784-786).  The reference does this with an
        all_gather per MICRO-batch (:709);
      * `SamplesProcessed` is published once per step (:805-808 publishes per micro-batch);
      * `maybe_send_weights()` applies the weight_update_interval rule after the optimizer step (:936-949);
      * resume: pass the `TrainingMetrics` restored from a checkpoint - samples, completed steps and the
        last broadcast version continue from there (:618-626)."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, rl_config: RLConfig, eos_token_id: int,
                 samples_per_step: int, max_train_steps: int, lr_scheduler: Any = None,
                 gradient_clipping_threshold: float | None = None, process_group: Any = None,
                 ref_model: torch.nn.Module | None = None, training_metrics: TrainingMetrics | None = None,
                 weight_update_manager: WeightUpdateManager | None = None, weight_update_interval: int = 1,
                 send_weight_updates: bool = True, trainer_stream: SingleStreamSpec | None = None,
                 equalize_micro_batches: bool = True, fused_ref_head: bool = True, skip_unlabelled: bool = True):
        """`ref_model`: a frozen reference policy on this GPU.  When given, the KL-to-reference term
        uses ITS log-probabilities, computed per micro-batch right before the policy forward (SURVEY
        §8f-3: replaces the HTTP round trip to a second inference server for KL-enabled configs).
        With `fused_ref_head` (default) a model in the Hugging Face layout (`.model` + bias-free `.lm_head`) is asked for
        its hidden states and the head runs on the MFMA kernels (no `[T, V]` reference logits); other models, or
        `fused_ref_head=False`, go through their logits and K1.
        `samples_per_step`: the GLOBAL number of samples per optimizer step (the loss normaliser).
        `equalize_micro_batches`: pad with sentinel micro-batches up to the largest count over the ranks
        (needed by ZeRO / FSDP; plain DDP only needs every rank to run at least one).
        `skip_unlabelled`: see `HotPathStep` - False keeps the reference's finiteness assert over every logits row (rl/__init__.py:213).
        An actor-critic model (`.value_head`, finetune/value_model.py; its forward also returns `.value` [B, L]) takes the value branch
        of `rl_step` per micro-batch (rl/__init__.py:162, 265-272, 367-381, 441-448; one `prl_value_head_fwd_bwd` launch): the advantages
        column of the step batch becomes rewards - V before the logits kernel reads it, the value loss's closed-form gradient goes
        back through `.value` next to d logits, and the five value statistics are accumulated over the step (`stats_dict`)."""
        import torch.distributed as dist

        self.has_value_head = getattr(_unwrap(model), "value_head", None) is not None
        self.skip_unlabelled = bool(skip_unlabelled)

        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.ref_model = ref_model
        self.fused_ref_head = bool(fused_ref_head)
        self.rl_config = rl_config.model_copy()
        self.rl_config.batch_size = samples_per_step
        self.samples_per_step = samples_per_step
        self.eos_token_id = eos_token_id
        self.max_train_steps = max_train_steps
        self.gradient_clipping_threshold = gradient_clipping_threshold
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.rank = dist.get_rank(process_group) if self.distributed else 0
        self.metrics = training_metrics or TrainingMetrics()
        self.weight_update_manager = weight_update_manager
        self.weight_update_interval = weight_update_interval
        self.send_weight_updates = send_weight_updates
        self.equalize = equalize_micro_batches
        self._last = None
        self._writer_cm = self._writer = None
        if trainer_stream is not None and self.rank == 0:
            self._writer_cm = write_to_streams(trainer_stream)
            self._writer = self._writer_cm.__enter__()

    def publish(self, message: BaseModel) -> None:
        if self._writer is not None:
            self._writer.write(message)

    def _share_counts(self, n_micro_batches: int, n_samples: int, device) -> tuple[int, int]:
        """(largest micro-batch count over the ranks, total samples) - one all-gather of two integers."""
        if not self.distributed or self.world == 1:
            return n_micro_batches, n_samples
        import torch.distributed as dist

        gloo = dist.get_backend(self.group) == "gloo"
        mine = torch.tensor([n_micro_batches, n_samples], dtype=torch.int64, device="cpu" if gloo else device)
        everyone = torch.empty((self.world, 2), dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(everyone, mine.unsqueeze(0), group=self.group)
        everyone = everyone.cpu()
        return int(everyone[:, 0].max()), int(everyone[:, 1].sum())

    def _sentinel_pass(self, device) -> None:
        """One zero-loss forward/backward (finetune_loop.py:784-786) so this rank takes part in the collectives."""
        from .finetune.utils import create_sentinel_batch

        b = create_sentinel_batch(device, tokenizer=type("T", (), {"eos_token_id": self.eos_token_id})(), model_version=0)
        out = self.model(input_ids=b.input_ids, attention_mask=b.attention_mask, position_ids=b.position_ids)
        if self.has_value_head:  # the critic's parameters take part in the reduction too
            torch.autograd.backward([out.logits, out.value], [torch.zeros_like(out.logits), torch.zeros_like(out.value)])
        else:
            out.logits.backward(torch.zeros_like(out.logits))

    def step(self, rollouts, micro_batches) -> dict[str, Any]:
        """`rollouts`: THIS rank's share of the step (ragged, on the device); `micro_batches`: its packing plan."""
        from .hotpath import HotPathStep

        hp = HotPathStep(self.rl_config, self.eos_token_id, self.metrics.completed_steps, self.max_train_steps, group=self.group,
                         skip_unlabelled=self.skip_unlabelled)
        batches = hp.preprocess(rollouts, micro_batches)
        n = len(batches)
        n_max, n_samples = self._share_counts(n, rollouts.n_seqs, rollouts.device)
        assert n_samples == self.samples_per_step, f"the ranks' shares hold {n_samples} samples, a step takes {self.samples_per_step}"
        n_passes = max(n_max, 1) if self.equalize else max(n, 1)
        vacc = None  # actor-critic: [value_mean, value_max, value_min, value_loss, value_mse] of the step so far (device, fp64)
        for j in range(n_passes):
            last = j == n_passes - 1
            ctx = self.model.no_sync() if (hasattr(self.model, "no_sync") and not last) else contextlib.nullcontext()
            if j >= n:  # this rank ran out of data: keep the collectives of the others company
                with ctx:
                    self._sentinel_pass(rollouts.device)
                continue
            b = batches[j]
            fused_ref = None
            if self.ref_model is not None and self.fused_ref_head:
                from .fused_head import _hidden_states, ref_head_for

                fused_ref = ref_head_for(self.ref_model)
            if fused_ref is not None:  # hidden states -> MFMA head: the [T, V] reference logits are never written
                with torch.no_grad():
                    hp.annotate_ref_logprobs_from_hidden(j, fused_ref[1], _hidden_states(fused_ref[0], b))
            elif self.ref_model is not None:
                with torch.no_grad():
                    ref_logits = self.ref_model(input_ids=b.input_ids, attention_mask=b.attention_mask, position_ids=b.position_ids).logits
                    if ref_logits.dtype not in (torch.float32, torch.bfloat16) or not ref_logits.is_contiguous():
                        ref_logits = ref_logits.float().contiguous()
                    hp.annotate_ref_logprobs(j, ref_logits)
                    del ref_logits
            with ctx:
                out = self.model(input_ids=b.input_ids, attention_mask=b.attention_mask, position_ids=b.position_ids)
                logits = out.logits
                g_val = None
                if self.has_value_head:
                    from .finetune.rl import value_head_terms

                    values = out.value
                    _, adv, vstats, g_val = value_head_terms(hp.cfg, b, values.detach(), want_grad=True)
                    # advantages := rewards - V (rl/__init__.py:272), IN the step batch: the logits kernel below and the step's ONE
                    # statistics launch (`finish`) read this column
                    # (`.data`: every column of the step batch is a view of ONE block, so an in-place write through the tensor itself would
                    # bump the version counter the model's saved `input_ids` share with it and autograd would refuse the backward)
                    b.advantages.data.copy_(adv.reshape(b.advantages.shape))
                    if vacc is None:
                        vacc = vstats.clone()
                    else:
                        vacc = torch.stack([vacc[0] + vstats[0], torch.maximum(vacc[1], vstats[1]), torch.minimum(vacc[2], vstats[2]),
                                            vacc[3] + vstats[3], vacc[4] + vstats[4]])
                lg = logits.detach()
                if lg.dtype not in (torch.float32, torch.bfloat16) or not lg.is_contiguous():
                    lg = lg.float().contiguous()
                dlogits = hp.logits_backward(j, lg)
                if g_val is not None:  # final_loss = policy_loss + value_loss_coef * value_loss (:381): both roots in one backward
                    torch.autograd.backward([logits, values], [dlogits.to(logits.dtype), (self.rl_config.value_loss_coef * g_val).to(values.dtype)])
                else:
                    logits.backward(dlogits.to(logits.dtype))
        loss, stats = hp.finish()
        value_stats = None
        if self.has_value_head:
            value_stats = self._reduce_value_stats(vacc if vacc is not None else torch.zeros(5, dtype=torch.float64, device=rollouts.device))
            loss = loss + self.rl_config.value_loss_coef * value_stats[3].to(loss.dtype)
        if self.gradient_clipping_threshold is not None:
            self.metrics.grad_norm = float(torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.gradient_clipping_threshold))
        self.optimizer.step()
        self.optimizer.zero_grad()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        m = self.metrics
        m.completed_steps += 1
        m.passes += n
        m.samples += self.samples_per_step
        m.tokens += int(hp.offsets[-1]) * self.world
        self.publish(SamplesProcessed(samples_processed=m.samples))
        self._last, self._last_value_stats = hp, value_stats
        return {"loss": loss, "stats": stats, "value_stats": value_stats, "micro_batches": n, "did_optimizer_step": True, "sentinel_passes": n_passes - n}

    def _reduce_value_stats(self, v: torch.Tensor) -> torch.Tensor:
        """The five value statistics over the data-parallel ranks: sums for the additive ones, max / min for the extrema."""
        if not self.distributed or self.world == 1:
            return v
        import torch.distributed as dist

        dev = v.device
        mine = v.cpu() if dist.get_backend(self.group) == "gloo" else v
        everyone = torch.empty((self.world, 5), dtype=v.dtype, device=mine.device)
        dist.all_gather_into_tensor(everyone, mine.unsqueeze(0).contiguous(), group=self.group)
        everyone = everyone.to(dev)
        return torch.stack([everyone[:, 0].sum(), everyone[:, 1].max(), everyone[:, 2].min(), everyone[:, 3].sum(), everyone[:, 4].sum()])

    def maybe_send_weights(self) -> bool:
        """After `step()`: broadcast when enough samples were trained since the last broadcast
        (`weight_update_interval`, reference :936-949).  Model version == samples trained."""
        m = self.metrics
        if not self.send_weight_updates or self.weight_update_manager is None:
            return False
        if m.samples - m.last_broadcasted_version < self.weight_update_interval:
            return False
        self.weight_update_manager.send_weight_update(m.samples)
        m.last_broadcasted_version = m.samples
        return True

    def finish(self) -> None:
        """Signal `TrainingDone` and close the trainer stream."""
        self.publish(TrainingDone())
        if self._writer_cm is not None:
            self._writer_cm.__exit__(None, None, None)
            self._writer_cm = self._writer = None

    def stats_dict(self, stats: torch.Tensor) -> dict[str, float]:
        """The step's statistics as the reference's dict; for an actor-critic model the reported loss is the combined one and the five
        value keys close the dict (rl/__init__.py:381-386, 441-448)."""
        out = self._last.stats_dict(stats)
        v = getattr(self, "_last_value_stats", None)
        if v is not None:
            from .finetune.rl import VALUE_STAT_KEYS

            vs = [float(np.float32(x)) for x in v.cpu().tolist()]
            combined = float(np.float32(out["loss"]) + np.float32(self.rl_config.value_loss_coef * np.float32(vs[VALUE_STAT_KEYS.index("value_loss")])))
            out["loss"] = out["max_loss"] = out["min_loss"] = combined
            out.update(dict(zip(VALUE_STAT_KEYS, vs)))
        return out
