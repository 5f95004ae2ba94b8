"""Tensor-parallel shard plan of a weight update.

The reference broadcasts every FULL tensor to every inference worker and lets each tensor-parallel rank
cut its own slice out of it inside `load_weights` (pipelinerl/vllm1.py:110-127): a TP = 2 engine receives
the parameter set twice and throws half of each copy away.  Here the trainer states, per parameter, along
which dimension the engine partitions it (`ParameterInfo.shard_dim` / `shard_parts` in the update request)
and every TP rank receives only its own slice, already in the shape its engine stores.

The rules below describe Megatron-style tensor parallelism as vLLM applies it to the Llama / Qwen family:

    embed_tokens, lm_head                      vocabulary-parallel        rows     (dim 0)
    q_proj, k_proj, v_proj, gate_proj, up_proj column-parallel            rows     (dim 0), their biases too
    o_proj, down_proj                          row-parallel               columns  (dim 1), bias replicated
    norms, everything else                     replicated

k_proj / v_proj of a grouped-query model with fewer KV heads than TP ranks are REPLICATED in groups: with
`parts` < tp_size, rank t holds part `t * parts // tp_size` (vLLM's `num_kv_head_replicas`).  Pass
`kv_heads=` to `plan_tp_shards` to get that; by default a dimension that does not divide by the TP degree
falls back to "replicated" rather than guessing.
"""

from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Iterable, Sequence

_COLUMN = re.compile(r"(^|\.)(q_proj|k_proj|v_proj|gate_proj|up_proj)\.(weight|bias)$")
_KV = re.compile(r"(^|\.)(k_proj|v_proj)\.(weight|bias)$")
_ROW = re.compile(r"(^|\.)(o_proj|down_proj)\.weight$")
_VOCAB = re.compile(r"(^|\.)(embed_tokens|lm_head)\.weight$")


@dataclass(frozen=True)
class TpShard:
    """How one parameter is cut: `dim` None = replicated (every rank gets the whole tensor), otherwise the
    tensor is split into `parts` equal pieces along `dim` and TP rank t takes piece `t * parts // tp_size`."""

    dim: int | None = None
    parts: int = 1

    def piece(self, tp_rank: int, tp_size: int) -> int:
        return 0 if self.dim is None else tp_rank * self.parts // tp_size

    def shard_shape(self, shape: Sequence[int]) -> tuple[int, ...]:
        if self.dim is None:
            return tuple(shape)
        s = list(shape)
        s[self.dim] //= self.parts
        return tuple(s)

    def bounds(self, shape: Sequence[int], tp_rank: int, tp_size: int) -> tuple[int, int]:
        """(start, length) along `dim` of this rank's piece."""
        n = shape[self.dim] // self.parts
        return self.piece(tp_rank, tp_size) * n, n


def default_tp_rule(name: str, shape: Sequence[int], tp_size: int, kv_heads: int | None = None) -> TpShard:
    """The table in the module docstring.  Anything that does not divide evenly stays replicated."""
    if tp_size <= 1:
        return TpShard()
    dim, parts = None, tp_size
    if _VOCAB.search(name) or _COLUMN.search(name):
        dim = 0
        if kv_heads is not None and _KV.search(name) and kv_heads < tp_size:
            if tp_size % kv_heads:
                return TpShard()
            parts = kv_heads
    elif _ROW.search(name):
        dim = 1
    if dim is None or dim >= len(shape) or shape[dim] % parts:
        return TpShard()
    return TpShard(dim, parts)


def plan_tp_shards(named_shapes: Iterable[tuple[str, Sequence[int]]], tp_size: int, kv_heads: int | None = None,
                   overrides: dict[str, TpShard] | None = None) -> dict[str, TpShard]:
    """name -> TpShard for a parameter list; `overrides` wins over the default rules (engine-specific layouts)."""
    out = {}
    for name, shape in named_shapes:
        out[name] = (overrides or {}).get(name) or default_tp_rule(name, shape, tp_size, kv_heads)
    return out


def shard_view(tensor, shard: TpShard, tp_rank: int, tp_size: int):
    """This rank's slice of a full tensor as a VIEW (dim 0: contiguous; dim 1: strided - the bucket gather
    makes it contiguous on the way into the staging buffer)."""
    if shard.dim is None:
        return tensor
    start, n = shard.bounds(tuple(tensor.shape), tp_rank, tp_size)
    return tensor.narrow(shard.dim, start, n)
