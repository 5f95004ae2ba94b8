"""`stateless_init_process_group` under the reference's name (pipelinerl/torch_utils.py:70-94): a communicator that is NOT the process's
default torch.distributed group - trainer rank 0 and the inference workers rendezvous over `tcp://host:port` - with the
`.broadcast(tensor, src, stream)` the reference's call sites use (finetune_loop.py:238, 282; vllm1.py:121).  Here it is a
`weight_sync.WeightSyncGroup` (RCCL over xGMI, which adds the bucketed forms) or, with `backend="gloo"`, its host-staged twin."""

from __future__ import annotations

from typing import Any

import torch


def stateless_init_process_group(init_method: str, rank: int, world_size: int, device: Any, backend: str = "rccl"):
    from .weight_sync import weight_sync_group

    return weight_sync_group(backend, init_method, rank, world_size, torch.device(device))
