"""Receive side of the weight update: a standalone shim with the method set of the reference's
vLLM `WorkerExtension` (pipelinerl/vllm1.py:62-134).  vLLM-ROCm is not part of this image, so the
shim is engine-agnostic: it needs `self.device`, `self.rank` and a `load_weights(list[(name,
tensor)]) -> iterable of loaded names` callable (vLLM's `model_runner.model.load_weights`).
Mix it into a vLLM worker class or use `StandaloneWeightReceiver` directly.
"""

from __future__ import annotations

import logging
from typing import Any

import torch

from .finetune_loop import WeightUpdateRequest
from .weight_sync import BucketedReceiver, WeightSyncGroup, string_to_dtype

logger = logging.getLogger(__name__)


class WorkerExtension:
    device: torch.device
    rank: int
    pg_rank: int
    model_update_group: Any

    def _load_weights(self, weights):  # overridden / provided by the engine
        return self.model_runner.model.load_weights(weights=weights)  # type: ignore[attr-defined]

    def _weight_destinations(self) -> dict[str, torch.Tensor] | None:
        """Optional: name -> the engine's own weight tensor for parameters that are stored exactly
        as the trainer names and shapes them.  Those are filled by one scatter-copy launch per bucket
        instead of going through `load_weights` (which stays in charge of fused / sharded weights)."""
        return None

    def _after_update(self) -> None:
        """Hook for engine caches that must be dropped after new weights land (the reference
        invalidates its fp32 lm_head cache here, vllm1.py:126)."""

    def init_actor_update_group(self, actor_idx: int, actor_ngpus: int, weight_update_group_init_method: str,
                                weight_update_group_world_size: int, tp_sharded: bool = False, backend: str = "rccl") -> None:
        """`tp_sharded`: join the communicator of THIS worker's tensor-parallel rank (trainer + the same TP rank of every
        engine) instead of the one spanning all workers: updates then arrive as this rank's slices only
        (`transport: sharded`, tp_shard.py).  The engine's TP degree is `actor_ngpus`."""
        # rank layout of the reference (vllm1.py:71): trainer = 0, worker = 1 + llm index * gpus + local rank
        self.pg_rank = 1 + actor_idx * actor_ngpus + self.rank
        logger.info(f"[INIT_ACTOR_UPDATE_GROUP]: actor {actor_idx}, ngpus {actor_ngpus}, rank {self.rank}, pg_rank {self.pg_rank}, "
                    f"init {weight_update_group_init_method}, world {weight_update_group_world_size}, tp_sharded {tp_sharded}")
        if tp_sharded:
            from .weight_sync import weight_sync_tp_groups

            self.model_update_group = weight_sync_tp_groups(backend, weight_update_group_init_method, self.pg_rank, weight_update_group_world_size,
                                                            actor_ngpus, self.device)[0]
            self.tp_rank, self.tp_size = self.rank, actor_ngpus
        else:
            from .weight_sync import weight_sync_group

            # backend "gloo": the same group without RCCL (hosts with one GPU, CPU tensors) - pipeline_run's weight_transport="gloo"
            self.model_update_group = weight_sync_group(backend, weight_update_group_init_method, self.pg_rank, weight_update_group_world_size, self.device)
        self._receiver = None

    def _load_weight_shards(self, shards):
        """`transport: sharded`: `shards` = [(name, tensor, TpShard)] where `tensor` is THIS TP rank's slice of the
        trainer's parameter `name` (already the shape a tensor-parallel engine stores; fused parameters such as
        qkv_proj / gate_up_proj take the slices of their parts side by side).  Engine specific: vLLM's `load_weights`
        expects full tensors and cannot be used here.  Register destinations (`_weight_shard_destinations`) or
        override this."""
        raise NotImplementedError("this worker has no loader for tensor-parallel slices: override _load_weight_shards "
                                  "or return the slices' storage from _weight_shard_destinations")

    def _weight_shard_destinations(self) -> dict[str, torch.Tensor] | None:
        """name (trainer side) -> the tensor that stores this TP rank's slice of it: filled straight from the
        received bucket by the scatter-copy kernel."""
        return None

    def receive_weight_update(self, request_json: str) -> None:
        request = WeightUpdateRequest.model_validate_json(request_json) if isinstance(request_json, str) else WeightUpdateRequest(**request_json)
        if torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)
        expected = (torch.bfloat16, torch.float32, torch.float16)
        for info in request.parameters_info:
            if string_to_dtype(info.dtype) not in expected:
                logger.warning(f"Unexpected dtype for {info.name}: {info.dtype}")

        def load(views):
            # vLLM's `load_weights` returns the ENGINE's parameter names: q/k/v_proj collapse into one
            # `qkv_proj`, gate/up_proj into `gate_up_proj`, so counts cannot be compared for a batch.
            # The reference passes one tensor per call and requires exactly one loaded name
            # (vllm1.py:120-124); for a bucket the same rule is applied tensor by tensor, which keeps
            # its unknown-parameter error exact.  Engines that can vouch for a whole batch implement
            # `_load_weights_batch` and return the trainer-side names they did not recognise.
            batch = getattr(self, "_load_weights_batch", None)
            if batch is not None and len(views) > 1:
                unknown = list(batch(views))
                if unknown:
                    raise ValueError(f"model {unknown} not found in model state dict")
                return
            for name, tensor in views:
                loaded = self._load_weights([(name, tensor)])
                if len(list(loaded)) != 1:
                    raise ValueError(f"model {name} not found in model state dict")

        if request.transport == "ipc":
            from .weight_sync import ColocatedReceiver

            if getattr(self, "_ipc_receiver", None) is None:
                self._ipc_receiver = ColocatedReceiver(self.device, request.bucket_bytes)
            self._ipc_receiver.receive([i.model_dump() for i in request.parameters_info], request.ipc_handles, request.ipc_nbytes, load,
                                       destinations=self._weight_destinations(), max_allocation=request.ipc_max_allocation)
        elif request.transport == "sharded":
            from .tp_shard import TpShard
            from .weight_sync import ParamSpec, plan_shard_buckets

            tp_rank, tp_size = getattr(self, "tp_rank", None), getattr(self, "tp_size", None)
            if tp_rank is None:
                raise RuntimeError("a sharded weight update arrived but init_actor_update_group was not called with tp_sharded=True")
            if tp_size != request.tp_size:
                raise ValueError(f"the update is cut for TP {request.tp_size}, this engine runs TP {tp_size}")
            specs = [ParamSpec(i.name, tuple(i.shape), string_to_dtype(i.dtype)) for i in request.parameters_info]
            cuts = {i.name: TpShard(i.shard_dim, i.shard_parts) for i in request.parameters_info}
            _, plan = plan_shard_buckets(specs, cuts, tp_rank, tp_size, request.bucket_bytes)
            if getattr(self, "_receiver", None) is None or self._receiver.bucket_bytes != request.bucket_bytes:
                self._receiver = BucketedReceiver(self.model_update_group, request.bucket_bytes)
            self._receiver.receive_planned(plan, lambda views: self._load_weight_shards([(n, t, cuts[n]) for n, t in views]),
                                           destinations=self._weight_shard_destinations())
        elif request.transport == "bucketed":
            if getattr(self, "_receiver", None) is None or self._receiver.bucket_bytes != request.bucket_bytes:
                self._receiver = BucketedReceiver(self.model_update_group, request.bucket_bytes)
            self._receiver.receive([i.model_dump() for i in request.parameters_info], load, destinations=self._weight_destinations())
        else:  # reference protocol: one broadcast per parameter
            for info in request.parameters_info:
                buf = torch.empty(tuple(info.shape), dtype=string_to_dtype(info.dtype), device=self.device)
                self.model_update_group.broadcast(buf, src=0)
                load([(info.name, buf)])
        self._after_update()
        logger.info("Weight update received")

    def close_communicator(self) -> None:
        ipc = getattr(self, "_ipc_receiver", None)
        if ipc is not None:
            ipc.close()
            self._ipc_receiver = None
        grp = getattr(self, "model_update_group", None)
        if grp is not None:
            grp.close()
            self.model_update_group = None
            logger.info("Weight update communicator closed")


class StandaloneWeightReceiver(WorkerExtension):
    """WorkerExtension over a plain `torch.nn.Module` (tests, non-vLLM engines): parameters are
    copied by name."""

    def __init__(self, module: torch.nn.Module, device: torch.device, rank: int = 0):
        self.module = module
        self.device = device
        self.rank = rank
        self._params = dict(module.named_parameters())

    def _weight_destinations(self):
        return {n: p.data for n, p in self._params.items()}

    def _load_weights(self, weights):
        loaded = []
        for name, tensor in weights:
            p = self._params.get(name)
            if p is None:
                continue
            p.data.copy_(tensor.to(p.dtype), non_blocking=True)
            loaded.append(name)
        return loaded

    def _load_weights_batch(self, weights):
        """One pass over a whole bucket; returns the names this module does not have."""
        done = set(self._load_weights(weights))
        return [n for n, _ in weights if n not in done]


class StandaloneShardReceiver(WorkerExtension):
    """One tensor-parallel rank of a plain-PyTorch "engine" (tests, non-vLLM engines): holds, for every trainer
    parameter, the slice `tp_shard.plan_tp_shards` assigns to `tp_rank`, and takes sharded updates straight into them."""

    def __init__(self, named_shapes, dtype_of, device: torch.device, tp_rank: int, tp_size: int, kv_heads: int | None = None):
        """`named_shapes`: [(name, full shape)], `dtype_of`: name -> dtype."""
        from .tp_shard import plan_tp_shards

        self.device = device
        self.rank = tp_rank
        self.cuts = plan_tp_shards(named_shapes, tp_size, kv_heads)
        self.slices = {n: torch.zeros(self.cuts[n].shard_shape(shape), dtype=dtype_of(n), device=device) for n, shape in named_shapes}

    def _weight_shard_destinations(self):
        return self.slices


class StackedShardReceiver(WorkerExtension):
    """One tensor-parallel rank of an engine that stores its weights the way vLLM does for the Llama / Qwen family:
    STACKED per layer - `qkv_proj` = [q rows of this rank; k rows; v rows] (`QKVParallelLinear`), `gate_up_proj` =
    [gate rows; up rows] (`MergedColumnParallelLinear`) - with `o_proj` / `down_proj` cut along their input dimension,
    `embed_tokens` / `lm_head` along the vocabulary and the norms replicated.  The reference gets there by handing
    every FULL tensor to `load_weights`, which cuts and stacks (vllm1.py:110-127); here
    `_weight_shard_destinations()` maps each TRAINER-side name to the VIEW of the stacked storage its slice belongs
    in, so a sharded update (`transport: sharded`) lands in place - no full tensors, no `load_weights`, no copy after
    the scatter kernel.

    With fewer KV heads than TP ranks the k / v rows are replicated in groups (`plan_tp_shards(kv_heads=...)`): the
    stacked tensor then has `q / tp + 2 * kv / kv_heads` rows, like vLLM's `num_kv_head_replicas` layout."""

    _STACKS = (("qkv_proj", ("q_proj", "k_proj", "v_proj")), ("gate_up_proj", ("gate_proj", "up_proj")))

    def __init__(self, named_shapes, dtype_of, device: torch.device, tp_rank: int, tp_size: int, kv_heads: int | None = None):
        from .tp_shard import plan_tp_shards

        self.device, self.rank = device, tp_rank
        named_shapes = [(n, tuple(s)) for n, s in named_shapes]
        self.cuts = plan_tp_shards(named_shapes, tp_size, kv_heads)
        shard_shape = {n: self.cuts[n].shard_shape(s) for n, s in named_shapes}
        self.storage: dict[str, torch.Tensor] = {}    # the engine's own (stacked) parameters
        self._views: dict[str, torch.Tensor] = {}     # trainer name -> view into `storage`
        by_name = dict(named_shapes)
        grouped: set[str] = set()
        for stacked, parts in self._STACKS:
            for n in by_name:
                head, _, tail = n.rpartition(f".{parts[0]}.")
                if not tail or n in grouped:
                    continue
                members = [f"{head}.{p}.{tail}" for p in parts]
                if not all(m in by_name for m in members):
                    continue
                rows = [shard_shape[m][0] for m in members]
                rest = shard_shape[members[0]][1:]
                buf = torch.zeros((sum(rows), *rest), dtype=dtype_of(members[0]), device=device)
                self.storage[f"{head}.{stacked}.{tail}"] = buf
                at = 0
                for m, r in zip(members, rows):
                    self._views[m] = buf.narrow(0, at, r)  # row blocks of a row-major tensor: contiguous views
                    at += r
                grouped.update(members)
        for n, _ in named_shapes:
            if n not in grouped:
                self.storage[n] = self._views[n] = torch.zeros(shard_shape[n], dtype=dtype_of(n), device=device)

    def _weight_shard_destinations(self):
        return self._views
