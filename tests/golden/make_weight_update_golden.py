"""Golden trace of the in-flight weight update, produced by EXECUTING the reference's own
sender and receiver with recording stubs around them:

  * sender: `WeightUpdateManager` (`pipelinerl/finetune_loop.py:174-292`), cut from the source and
    `exec`ed.  Stand-ins: `deepspeed` / `FSDP` / `AutoModelForCausalLMWithValueHead` (classes the plain
    model is not an instance of), `get_accelerator()` (main process, identity unwrap), `requests`
    (records the POST instead of sending it), `torch.cuda.current_stream` (no GPU here).
  * receiver: `WorkerExtension` (`pipelinerl/vllm1.py:62-134`), cut from the source and `exec`ed.
    Stand-ins: `stateless_init_process_group` (records its arguments), `torch.cuda.synchronize` /
    `current_stream`, `pipelinerl.vllm_quantization.invalidate_fp32_cache` (records the call);
    `string_to_dtype` is cut from `vllm_quantization.py:64-82` and executed as well.
  * the "process group" is a loop-back object: what the sender broadcasts the receiver's broadcast
    copies into its buffer, in order.
  * message models, streams: as in make_trainer_golden.py (reference code, executed).

Recorded: the HTTP request (URL + JSON body), the order / shape / dtype / source rank of every
broadcast, the stream record published afterwards, every `load_weights` call on the worker with a
checksum of what it received, the cache-invalidation call, the error raised for an unknown
parameter name, the process-group rank arithmetic of `init_actor_update_group`, and the
`string_to_dtype` table.  Nothing of the reference is stored, only this trace
(tests/golden/weight_update_trace.json).

    python tests/golden/make_weight_update_golden.py
"""

from __future__ import annotations

import json
import logging
import os
import sys
import tempfile
import textwrap
import types
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
REF = Path("/root/reference/pipelinerl")


def cut(path: Path, start_prefix: str, end_prefix: str) -> str:
    lines = path.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(start_prefix))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(end_prefix))
    return textwrap.dedent("\n".join(lines[start:end]))


def toy_model() -> torch.nn.Module:
    """Same construction as tests/test_weight_update_golden.py (seeded)."""
    torch.manual_seed(7)
    m = torch.nn.Sequential()
    m.add_module("embed", torch.nn.Embedding(11, 6))
    m.add_module("proj", torch.nn.Linear(6, 5))
    m.add_module("norm", torch.nn.LayerNorm(5))
    m = m.to(torch.bfloat16)
    m.norm = m.norm.to(torch.float32)
    return m


class Loopback:
    """Stands for the NCCL group: the trainer's broadcasts feed the worker's."""

    def __init__(self):
        self.sent, self.log = [], []

    def sender(self):
        outer = self

        class S:
            def broadcast(self, tensor, src, stream=None):
                outer.sent.append(tensor.detach().clone())
                outer.log.append({"side": "trainer", "shape": list(tensor.shape), "dtype": str(tensor.dtype), "src": src})

        return S()

    def receiver(self):
        outer = self

        class R:
            def broadcast(self, tensor, src, stream=None):
                tensor.copy_(outer.sent.pop(0))
                outer.log.append({"side": "worker", "shape": list(tensor.shape), "dtype": str(tensor.dtype), "src": src})

        return R()


def main() -> None:
    from make_trainer_golden import load_reference

    m, _state, streams = load_reference()
    streams.set_streams_backend("files")
    cuda_shim = types.SimpleNamespace(current_stream=lambda *a, **k: None, synchronize=lambda *a, **k: None)
    torch_shim = types.SimpleNamespace(cuda=cuda_shim, empty=torch.empty, bfloat16=torch.bfloat16, float32=torch.float32,
                                       float16=torch.float16, dtype=torch.dtype, device=torch.device)

    # ---------------- sender ----------------
    posts = []

    class Response:
        status_code, text = 200, "ok"

        def raise_for_status(self):
            pass

    requests = types.SimpleNamespace(post=lambda url, json=None: (posts.append({"url": url, "json": json}), Response())[1],
                                     RequestException=Exception)
    accel = types.SimpleNamespace(is_main_process=True, unwrap_model=lambda x: x, wait_for_everyone=lambda: None)
    ns = dict(m.__dict__)
    ns.update(
        ThreadPoolExecutor=ThreadPoolExecutor, requests=requests, logger=logging.getLogger("ref"), torch=torch_shim,
        deepspeed=types.SimpleNamespace(DeepSpeedEngine=type("DeepSpeedEngine", (), {})),
        AutoModelForCausalLMWithValueHead=type("AutoModelForCausalLMWithValueHead", (), {}),
        FSDP=type("FSDP", (), {}), get_accelerator=lambda: accel, write_to_streams=streams.write_to_streams,
    )
    exec(compile(cut(REF / "finetune_loop.py", "class WeightUpdateManager", "def get_batch_token_count"), "ref_sender", "exec"), ns)
    model = toy_model()
    link = Loopback()
    with tempfile.TemporaryDirectory() as td:
        exp = Path(td)
        spec = streams.SingleStreamSpec(exp_path=exp, topic=m.TRAINER_TOPIC)
        mgr = ns["WeightUpdateManager"](["http://llm0:8080", "http://llm1:8081"], model, spec, link.sender())
        mgr.send_weight_update(5)
        mgr.shutdown()
        stream_text = next(exp.rglob("*.jsonl")).read_text()
    stream_records = [json.loads(l) for l in stream_text.splitlines()]
    for r in stream_records:
        r.pop("timestamp", None)
    for p in posts:
        p["json"].pop("timestamp", None)
    assert posts[0]["json"] == posts[1]["json"]

    # ---------------- receiver ----------------
    exec(compile(cut(REF / "vllm_quantization.py", "def string_to_dtype", "def _resolve_dtype_from_config"), "ref_s2d", "exec"),
         s2d_ns := {"torch": torch})
    string_to_dtype = s2d_ns["string_to_dtype"]
    invalidations = []
    pg_calls = []
    wns = dict(
        WeightUpdateRequest=m.WeightUpdateRequest, torch=torch_shim, logger=logging.getLogger("ref"), os=os,
        string_to_dtype=string_to_dtype, LikeWorker=object, Any=object,
        stateless_init_process_group=lambda **kw: (pg_calls.append({k: str(v) for k, v in kw.items()}), "group")[1],
        pipelinerl=types.SimpleNamespace(vllm_quantization=types.SimpleNamespace(invalidate_fp32_cache=lambda: invalidations.append(True))),
    )
    exec(compile(cut(REF / "vllm1.py", "class WorkerExtension", "class WeightUpdateManager"), "ref_receiver", "exec"), wns)
    loads = []

    class Engine(wns["WorkerExtension"]):
        def __init__(self, known):
            self.device, self.rank = torch.device("cpu"), 0
            self.model_update_group = link.receiver()
            engine = self

            class Model:
                def load_weights(self, weights):
                    out = set()
                    for name, t in weights:
                        if name in known:
                            loads.append({"name": name, "shape": list(t.shape), "dtype": str(t.dtype),
                                          "sum": float(t.double().sum()), "abs_sum": float(t.double().abs().sum())})
                            out.add(name)
                    return out

            self.model_runner = types.SimpleNamespace(model=Model())

    request_json = json.dumps({**posts[0]["json"], "timestamp": 1.0})
    names = [p["name"] for p in posts[0]["json"]["parameters_info"]]
    Engine(set(names)).receive_weight_update(request_json)
    n_invalidations_ok = len(invalidations)
    n_loads_ok, n_log_ok = len(loads), len(link.log)
    # unknown parameter on the worker: resend, the engine does not know the second name
    link.sent = [p.detach().clone() for _, p in model.named_parameters()]
    err = None
    try:
        Engine(set(names) - {names[1]}).receive_weight_update(request_json)
    except ValueError as e:
        err = {"type": "ValueError", "mentions_name": names[1] in str(e)}
    # rank arithmetic
    ranks = []
    for actor_idx, ngpus, rank in ((0, 1, 0), (1, 1, 0), (0, 2, 1), (1, 2, 0), (1, 2, 1), (3, 4, 2)):
        e = Engine(set())
        e.rank = rank
        e.init_actor_update_group(actor_idx, ngpus, "tcp://127.0.0.1:9000", 9)
        ranks.append({"actor_idx": actor_idx, "actor_ngpus": ngpus, "rank": rank, "pg_rank": e.pg_rank, "init_call": pg_calls[-1]})
    e.close_communicator()
    dtypes = {}
    for s in ("torch.bfloat16", "bfloat16", "bf16", "torch.float32", "fp32", "float", "torch.float16", "half", "BF16", " torch.Float32 ", "int8", "torch.int64"):
        try:
            dtypes[s] = str(string_to_dtype(s))
        except ValueError:
            dtypes[s] = "ValueError"
    out = {
        "llm_urls": ["http://llm0:8080", "http://llm1:8081"], "version": 5,
        "posts": posts, "link_log": link.log[:n_log_ok], "stream_records": stream_records, "loads": loads[:n_loads_ok],
        "unknown_parameter_loads_before_error": [l["name"] for l in loads[n_loads_ok:]],
        "cache_invalidations_total": len(invalidations),
        "cache_invalidations_after_success": n_invalidations_ok, "unknown_parameter": err,
        "close_sets_group_none": e.model_update_group is None, "pg_ranks": ranks, "string_to_dtype": dtypes,
    }
    (HERE / "weight_update_trace.json").write_text(json.dumps(out, indent=1))
    print(json.dumps({k: out[k] for k in ("posts", "stream_records", "unknown_parameter", "string_to_dtype")}, indent=1)[:3000])
    print(len(link.log), "broadcast events;", len(loads), "load_weights calls")


if __name__ == "__main__":
    main()
