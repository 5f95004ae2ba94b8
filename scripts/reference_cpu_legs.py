"""Build-container measurement of the REFERENCE's own functions on host cores, leg by leg as BASELINE.md §2 lists
them - the constants bench.py prints beside the port's same legs measured on the GPU box.  Reads /root/reference, so it
only runs in the build container (the GPU box has no reference); the result is committed under profiles/.

    python scripts/reference_cpu_legs.py [--threads N] [--out profiles/r03_reference_cpu_legs.json]

Legs (same seeded synthetic rollouts as bench.py's `cpu_baseline`: 64 dense 8192-token sequences):
  preprocess      `preprocess_fn` x 64 + `populate_rl_data`          (finetune/data.py:111, rl/__init__.py:453)
  collate_packed  one sequence per 8192-token micro-batch x 64       (finetune/data.py:215)
  wire            files-backend record round trip of those micro-batches.  `pipelinerl/streams.py` itself needs orjson
                  and redis (absent, no network): the leg follows its format - `json.dumps(batch.model_dump())` + "\\n",
                  `json.loads` + `PipelineBatchEncoding(**d)` (streams.py:249-346, types.py:77-110) - with the stdlib
                  encoder, and says so.
  loss_v8         `rl_step`, fake logits V = 8, T = 2048: token loss + reduce + 32 statistics alone (K2 + K3)
  logprob_fwd / logprob_bwd   `rl_step`, V = 152 064, T = 2048: forward, and autograd backward to the logits (K1)
Median of 5 after 2 warm-ups for the cheap legs, median of 3 after 1 for the V = 152 064 legs (one pass is seconds)."""
import argparse
import json
import os
import statistics
import sys
import time
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
ap.add_argument("--seqs", type=int, default=64)
ap.add_argument("--seq-length", type=int, default=8192)
ap.add_argument("--vocab", type=int, default=152064)
ap.add_argument("--tokens", type=int, default=2048)
ap.add_argument("--out", default=str(ROOT / "profiles" / "r03_reference_cpu_legs.json"))
args = ap.parse_args()

sys.path.insert(0, "/root/reference")
om = types.ModuleType("omegaconf")
om.DictConfig = om.ListConfig = om.OmegaConf = object
sys.modules.setdefault("omegaconf", om)
from pipelinerl.finetune.data import collate_packed, preprocess_fn  # noqa: E402
from pipelinerl.finetune.rl import RLConfig, populate_rl_data, rl_step  # noqa: E402
from pipelinerl.finetune.types import PipelineBatchEncoding  # noqa: E402

from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries  # noqa: E402  (the shared synthetic generator)

torch.set_num_threads(args.threads)


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


class Tok:
    eos_token_id, padding_side = 2, "right"


rag, reasons = make_ragged(args.seqs // 8, attempts=8, seq_length=args.seq_length, vocab=args.vocab, seed=99, dense=True)
entries = ragged_to_entries(rag, reasons)
n_tok = sum(len(e["input_ids"]) for e in entries)
cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, clamp_log_ratio_ref_new_value=5,
               divide_advantage_by_std=False, batch_size=4096, temperature=1.0)
legs = {}


def preprocess():
    data = [preprocess_fn(dict(e), Tok(), args.seq_length, is_rl=True) for e in entries]
    for d, e in zip(data, entries):  # what preprocess.py:152-175 carries over next to the encoding
        d.update(group_id=e["group_id"], rollout_index=e["metadata"]["rollout_index"], step_index=e["metadata"]["step_index"],
                 model_version=e["metadata"]["model_version"], finished=e["finished"])
    return populate_rl_data(data, Tok.eos_token_id, cfg)


t = timed(preprocess, 5, 2)
legs["preprocess"] = {"s": t, "us_per_token": 1e6 * t / n_tok, "samples_per_s": args.seqs / t, "what": "preprocess_fn x %d + populate_rl_data" % args.seqs}
data = preprocess()
t = timed(lambda: [collate_packed([d], Tok(), 1) for d in data], 5, 2)
legs["collate_packed"] = {"s": t, "us_per_token": 1e6 * t / n_tok, "what": "collate_packed, one sequence per micro-batch x %d" % args.seqs}
batches = [collate_packed([d], Tok(), 1) for d in data]


def to_jsonable(b):
    d = b.model_dump()
    return {k: (v.tolist() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


wire_n = 16  # of the 64 micro-batches: the text path is slow
sizes = []


def wire():
    sizes.clear()
    for b in batches[:wire_n]:
        line = json.dumps(to_jsonable(b)) + "\n"
        sizes.append(len(line))
        PipelineBatchEncoding(**json.loads(line))


t = timed(wire, 3, 1)
wire_tok = sum(int(b.input_ids.numel()) for b in batches[:wire_n])
legs["wire"] = {"s": t, "us_per_token": 1e6 * t / wire_tok, "bytes_per_token": sum(sizes) / wire_tok, "micro_batches": wire_n,
                "what": "files-backend record format (streams.py:249-346: one JSON object per line, tensors as nested lists) encode + decode + "
                        "list->tensor coercion (types.py:77-110); stdlib json stands in for orjson (absent)"}

T = args.tokens
b0 = batches[0]
sl = {k: (v[:, :T].contiguous() if isinstance(v, torch.Tensor) and v.dim() == 2 else v) for k, v in b0.model_dump().items()}
sl["seq_boundaries"] = torch.tensor([0, T], dtype=torch.int32)
batch = PipelineBatchEncoding(**sl)
rng = np.random.default_rng(0)
for name, V, reps, warm in (("loss_v8", 8, 5, 2), ("logprob", args.vocab, 3, 1)):
    logits = torch.nn.Parameter(torch.from_numpy((rng.standard_normal((1, T, V)) * 2).astype(np.float32)))
    model = lambda **kw: types.SimpleNamespace(logits=logits)  # noqa: E731
    if V == 8:
        batch.input_ids = batch.input_ids % 8

    fwd_t, bwd_t = [], []
    for it in range(reps + warm):
        logits.grad = None
        t0 = time.perf_counter()
        loss, stats = rl_step(model, batch, 0, 10, cfg)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        if it >= warm:
            fwd_t.append(t1 - t0)
            bwd_t.append(t2 - t1)
    f, bw = statistics.median(fwd_t), statistics.median(bwd_t)
    if V == 8:
        legs["loss_v8"] = {"s_fwd": f, "s_bwd": bw, "us_per_token": 1e6 * f / T, "us_per_token_fwd_bwd": 1e6 * (f + bw) / T, "tokens": T,
                           "what": "rl_step with V = 8 fake logits: token loss + reduce + 32 statistics (K2 + K3), forward; backward beside it"}
    else:
        legs["logprob_fwd"] = {"s": f, "us_per_token": 1e6 * f / T, "tokens": T, "vocab": V, "what": "rl_step forward, V = 152 064 (K1 + K1e + K2 + K3)"}
        legs["logprob_bwd"] = {"s": bw, "us_per_token": 1e6 * bw / T, "tokens": T, "vocab": V, "what": "autograd backward of the same call to the logits"}
    del logits

model_name = next((line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name")), "?")
per_sample = (legs["preprocess"]["s"] + legs["collate_packed"]["s"]) / args.seqs + (legs["logprob_fwd"]["us_per_token"] + legs["logprob_bwd"]["us_per_token"]) * 1e-6 * args.seq_length
out = {
    "what": "the reference's own functions (imported from /root/reference) on host cores, BASELINE.md §2 legs",
    "threads": args.threads, "host": {"nproc": os.cpu_count(), "model": model_name},
    "workload": {"sequences": args.seqs, "seq_length": args.seq_length, "tokens": n_tok, "vocab": args.vocab, "loss_tokens": T},
    "legs": legs,
    "samples_per_s_extrapolated": 1.0 / per_sample,
    "measured_in": "build container (reads /root/reference); constants in bench.py, not re-measured on the GPU box",
}
Path(args.out).write_text(json.dumps(out) + "\n")
print(json.dumps(out, indent=1))
