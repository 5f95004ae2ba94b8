"""The weight-update protocol against a trace of the reference's own sender and receiver
(tests/golden/make_weight_update_golden.py executes `WeightUpdateManager.send_weight_update`,
finetune_loop.py:174-292, and `WorkerExtension`, vllm1.py:62-134, with recording stubs).

With `transport="per_tensor"` this package must put exactly the reference's traffic on the wire:
same HTTP requests, same broadcasts in the same order, same stream record, same `load_weights`
calls on the worker, same errors.  CPU only (host tensors through a loop-back group)."""

import json

import pytest
import torch

from helpers import GOLDEN


@pytest.fixture(scope="module")
def g():
    return json.loads((GOLDEN / "weight_update_trace.json").read_text())


def toy_model() -> torch.nn.Module:
    """Same construction as make_weight_update_golden.toy_model (seeded)."""
    torch.manual_seed(7)
    m = torch.nn.Sequential()
    m.add_module("embed", torch.nn.Embedding(11, 6))
    m.add_module("proj", torch.nn.Linear(6, 5))
    m.add_module("norm", torch.nn.LayerNorm(5))
    m = m.to(torch.bfloat16)
    m.norm = m.norm.to(torch.float32)
    return m


class Loopback:
    def __init__(self):
        self.sent, self.log = [], []

    def side(self, name, receive):
        outer = self

        class Group:
            device = torch.device("cpu")

            def broadcast(self, tensor, src, stream=None):
                if receive:
                    tensor.copy_(outer.sent.pop(0))
                else:
                    outer.sent.append(tensor.detach().clone())
                outer.log.append({"side": name, "shape": list(tensor.shape), "dtype": str(tensor.dtype), "src": src})

            def close(self):
                pass

        return Group()


def _engine(known, link, loads):
    from pipelinerl_amd.vllm_worker import WorkerExtension

    class Engine(WorkerExtension):
        def __init__(self):
            self.device, self.rank = torch.device("cpu"), 0
            self.model_update_group = link.side("worker", True)
            self.invalidations = 0

        def _load_weights(self, weights):
            out = set()
            for name, t in weights:
                if name in known:
                    loads.append({"name": name, "shape": list(t.shape), "dtype": str(t.dtype), "sum": float(t.double().sum()),
                                  "abs_sum": float(t.double().abs().sum())})
                    out.add(name)
            return out

        def _after_update(self):
            self.invalidations += 1

    return Engine()


def test_per_tensor_protocol_is_the_reference_traffic(g, tmp_path):
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune_loop import TRAINER_TOPIC, WeightUpdateManager

    posts = []
    link = Loopback()
    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        spec = streams.SingleStreamSpec(exp_path=tmp_path, topic=TRAINER_TOPIC)
        mgr = WeightUpdateManager(g["llm_urls"], toy_model(), spec, link.side("trainer", False), transport="per_tensor",
                                  post=lambda url, payload: posts.append({"url": url, "json": payload}))
        mgr.send_weight_update(g["version"])
        mgr.shutdown()
        records = [json.loads(l) for l in next(tmp_path.rglob("*.jsonl")).read_text().splitlines()]
    finally:
        streams.reset_streams_backend()
    # HTTP: one POST per server, the reference's URL and body (+ the transport extension fields)
    assert sorted(p["url"] for p in posts) == sorted(p["url"] for p in g["posts"])
    for p in posts:
        body = dict(p["json"])
        body.pop("timestamp")
        want = g["posts"][0]["json"]
        assert {k: body[k] for k in want} == want
        assert body["transport"] == "per_tensor" and set(body) - set(want) <= {"transport", "bucket_bytes", "ipc_handles", "ipc_nbytes", "ipc_max_allocation", "tp_size"}
    # stream: WeightUpdateSuccess after the broadcasts
    for r in records:
        r.pop("timestamp")
    assert records == g["stream_records"]
    # wire: same tensors in the same order from rank 0
    trainer_log = [e for e in g["link_log"] if e["side"] == "trainer"]
    assert link.log == trainer_log

    # worker side: the reference's request JSON drives this package's receiver
    loads = []
    eng = _engine({p["name"] for p in g["posts"][0]["json"]["parameters_info"]}, link, loads)
    eng.receive_weight_update(json.dumps({**g["posts"][0]["json"], "timestamp": 1.0}))
    assert [e for e in link.log if e["side"] == "worker"] == [e for e in g["link_log"] if e["side"] == "worker"]
    assert loads == g["loads"]  # names, shapes, dtypes and the exact values that arrived
    assert eng.invalidations == g["cache_invalidations_after_success"]
    eng.close_communicator()
    assert (eng.model_update_group is None) == g["close_sets_group_none"]


def test_unknown_parameter_raises_like_the_reference(g):
    link, loads = Loopback(), []
    link.sent = [p.detach().clone() for _, p in toy_model().named_parameters()]
    names = [p["name"] for p in g["posts"][0]["json"]["parameters_info"]]
    eng = _engine(set(names) - {names[1]}, link, loads)
    with pytest.raises(ValueError) as e:
        eng.receive_weight_update(json.dumps({**g["posts"][0]["json"], "timestamp": 1.0}))
    assert g["unknown_parameter"]["type"] == "ValueError" and (names[1] in str(e.value)) == g["unknown_parameter"]["mentions_name"]
    assert [l["name"] for l in loads] == g["unknown_parameter_loads_before_error"]
    assert eng.invalidations == g["cache_invalidations_total"] - g["cache_invalidations_after_success"]  # none on failure


def test_pg_rank_arithmetic_and_dtype_table(g, monkeypatch):
    from pipelinerl_amd import vllm_worker
    from pipelinerl_amd.weight_sync import string_to_dtype

    calls = []
    monkeypatch.setattr(vllm_worker.WeightSyncGroup, "from_init_method",
                        classmethod(lambda cls, init_method, rank, world_size, device, timeout_s=300.0: calls.append(
                            {"init_method": str(init_method), "rank": str(rank), "world_size": str(world_size), "device": str(device)}) or "group"))
    for rec in g["pg_ranks"]:
        eng = _engine(set(), Loopback(), [])
        eng.rank = rec["rank"]
        eng.init_actor_update_group(rec["actor_idx"], rec["actor_ngpus"], rec["init_call"]["init_method"], int(rec["init_call"]["world_size"]))
        assert eng.pg_rank == rec["pg_rank"] and calls[-1] == rec["init_call"]
    for s, want in g["string_to_dtype"].items():
        if want == "ValueError":
            with pytest.raises(ValueError):
                string_to_dtype(s)
        else:
            assert str(string_to_dtype(s)) == want, s
