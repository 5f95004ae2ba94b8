"""The receive shim against an engine that behaves like vLLM's `load_weights`: trainer-side names
collapse into the engine's stacked parameters (q/k/v_proj -> qkv_proj, gate/up_proj -> gate_up_proj)
and the call returns ENGINE names.  Bucketed transport (the default of WeightUpdateManager) must
work against it and still raise the reference's unknown-parameter error (vllm1.py:120-124)."""

import json

import pytest
import torch

from pipelinerl_amd.finetune_loop import ParameterInfo, WeightUpdateRequest
from pipelinerl_amd.vllm_worker import WorkerExtension
from pipelinerl_amd.weight_sync import BucketedSender

STACKED = {"q_proj": "qkv_proj", "k_proj": "qkv_proj", "v_proj": "qkv_proj", "gate_proj": "gate_up_proj", "up_proj": "gate_up_proj"}


class LoopGroup:
    """Both ends of the update group in one process: the sender's buckets are replayed to the receiver."""

    device = torch.device("cpu")

    def __init__(self):
        self.sent: list[torch.Tensor] = []
        self.receiving = False

    def broadcast_bucket(self, buf, mode="scatter_allgather", src=0):
        if self.receiving:
            buf.copy_(self.sent.pop(0))
        else:
            self.sent.append(buf.clone())

    def close(self):
        pass


class VllmLikeEngine(WorkerExtension):
    def __init__(self, group, known_engine_params):
        self.device, self.rank = torch.device("cpu"), 0
        self.model_update_group = group
        self.known = known_engine_params
        self.loaded: dict[str, torch.Tensor] = {}

    def _load_weights(self, weights):
        out = set()
        for name, t in weights:
            parts = name.split(".")
            engine_name = ".".join(parts[:-2] + [STACKED.get(parts[-2], parts[-2]), parts[-1]])
            if engine_name in self.known:
                self.loaded[name] = t.clone()
                out.add(engine_name)
        return out


def _params():
    torch.manual_seed(0)
    names = [f"model.layers.0.self_attn.{p}.weight" for p in ("q_proj", "k_proj", "v_proj", "o_proj")]
    names += [f"model.layers.0.mlp.{p}.weight" for p in ("gate_proj", "up_proj", "down_proj")] + ["lm_head.weight"]
    return [(n, torch.randn(5, 3).bfloat16()) for n in names]


def _engine_names(params):
    out = set()
    for n, _ in params:
        parts = n.split(".")
        out.add(".".join(parts[:-2] + [STACKED.get(parts[-2], parts[-2]), parts[-1]]))
    return out


@pytest.mark.parametrize("bucket_bytes", [64, 1 << 20])
def test_bucketed_update_into_an_engine_with_stacked_parameters(bucket_bytes):
    params = _params()
    group = LoopGroup()
    BucketedSender(group, bucket_bytes=bucket_bytes).send(params)
    group.receiving = True
    eng = VllmLikeEngine(group, _engine_names(params))
    req = WeightUpdateRequest(version=1, transport="bucketed", bucket_bytes=bucket_bytes,
                              parameters_info=[ParameterInfo(name=n, shape=list(p.shape), dtype=str(p.dtype)) for n, p in params])
    eng.receive_weight_update(req.model_dump_json())
    assert set(eng.loaded) == {n for n, _ in params}
    for n, p in params:
        assert torch.equal(eng.loaded[n], p), n


def test_bucketed_update_reports_the_unknown_parameter():
    params = _params()
    group = LoopGroup()
    BucketedSender(group, bucket_bytes=1 << 20).send(params)
    group.receiving = True
    eng = VllmLikeEngine(group, _engine_names(params) - {"lm_head.weight"})
    req = WeightUpdateRequest(version=1, transport="bucketed", bucket_bytes=1 << 20,
                              parameters_info=[ParameterInfo(name=n, shape=list(p.shape), dtype=str(p.dtype)) for n, p in params])
    with pytest.raises(ValueError, match="lm_head.weight"):
        eng.receive_weight_update(json.loads(req.model_dump_json()))
