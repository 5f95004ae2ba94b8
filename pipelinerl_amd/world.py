"""Single-node actor / preprocessor / learner GPU partition (the part of reference
pipelinerl/world.py:143-192 the hot path depends on).  Multi-node placement, DNS naming and job
maps are orchestration and out of scope."""

from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class GpuPartition:
    actor_gpus: list[int]
    preprocessor_gpus: list[int]
    finetune_gpus: list[int]
    llms_per_actor: int
    gpus_per_llm: int
    total_actor_llms: int
    weight_update_group_size: int  # trainer rank 0 + every inference-worker GPU (world.py:192)

    @property
    def total_finetune_gpus(self) -> int:
        return len(self.finetune_gpus)


def split_gpus(
    total_gpus: int,
    actor_fraction: float = 4,
    preprocessor_fraction: float = 0,
    finetune_fraction: float = 4,
    tensor_parallel_size: int = 1,
    pipeline_parallel_size: int = 1,
    replicas: int = 1,
) -> GpuPartition:
    """Actors take `int(total * af / (af + pf + ff))` GPUs (at least one LLM's worth) rounded down
    to whole LLM instances per replica, likewise the reference-model servers of the preprocessor
    when its fraction is non-zero; the learners take what is left.  Defaults are conf/base.yaml's
    4 : 0 : 4."""
    gpus_per_llm = tensor_parallel_size * pipeline_parallel_size
    fsum = actor_fraction + preprocessor_fraction + finetune_fraction
    want_actor = max(int(total_gpus * actor_fraction / fsum), gpus_per_llm)
    want_pre = max(int(total_gpus * preprocessor_fraction / fsum), gpus_per_llm) if preprocessor_fraction else 0

    def per_replica(share: int) -> int:
        g = int(share / replicas) if replicas > 0 else 0
        return g - (g % gpus_per_llm)

    gpus_per_actor = per_replica(want_actor)
    gpus_per_pre = per_replica(want_pre)
    llms_per_actor = max(int(gpus_per_actor / gpus_per_llm), 1) if gpus_per_actor > 0 else 0
    total_actor_gpus = replicas * gpus_per_actor
    total_pre_gpus = replicas * gpus_per_pre
    n_finetune = total_gpus - total_actor_gpus - total_pre_gpus
    if n_finetune < 0:
        raise ValueError("Not enough gpus to place all workers")
    total_actor_llms = llms_per_actor * replicas
    ids = list(range(total_gpus))
    return GpuPartition(
        actor_gpus=ids[:total_actor_gpus],
        preprocessor_gpus=ids[total_actor_gpus : total_actor_gpus + total_pre_gpus],
        finetune_gpus=ids[total_actor_gpus + total_pre_gpus :],
        llms_per_actor=llms_per_actor,
        gpus_per_llm=gpus_per_llm,
        total_actor_llms=total_actor_llms,
        weight_update_group_size=total_actor_llms * gpus_per_llm + 1,
    )


def round_up_accumulation_passes(gradient_accumulation_passes: int, total_finetune_gpus: int) -> int:
    """The launcher rounds the accumulation passes up to a multiple of the learner count
    (reference launch.py:631-640) so that every rank gets whole micro-batches."""
    if total_finetune_gpus <= 0:
        return gradient_accumulation_passes
    r = gradient_accumulation_passes % total_finetune_gpus
    return gradient_accumulation_passes if r == 0 else gradient_accumulation_passes + total_finetune_gpus - r
