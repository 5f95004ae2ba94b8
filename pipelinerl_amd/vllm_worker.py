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
                                weight_update_group_world_size: int) -> None:
        # rank layout of the reference (vllm1.py:71): trainer = 0, worker = 1 + llm index * gpus + local rank
        self.pg_rank = 1 + actor_idx * actor_ngpus + self.rank
        logger.info(f"[INIT_ACTOR_UPDATE_GROUP]: actor {actor_idx}, ngpus {actor_ngpus}, rank {self.rank}, pg_rank {self.pg_rank}, "
                    f"init {weight_update_group_init_method}, world {weight_update_group_world_size}")
        self.model_update_group = WeightSyncGroup.from_init_method(
            weight_update_group_init_method, rank=self.pg_rank, world_size=weight_update_group_world_size, device=self.device
        )
        self._receiver = None

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
                                       destinations=self._weight_destinations())
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
