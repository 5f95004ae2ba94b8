"""Benchmark of the learner hot path on MI355X (contract: see the round prompt / DESIGN.md §5).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one optimizer step's worth of the post-model hot path for the BASELINE config
"Qwen2.5-7B GRPO bs=4096 seq=8192" on synthetic rollouts already resident in HBM:

    K5  group advantages over the rank's sequences
    K6  ONE pack launch writing every micro-batch of the step (PipelineBatchEncoding layout)
    per micro-batch (8192 tokens, V = 152 064 fp32 logits resident in HBM):
        fused K1 + token gradient + K1 backward  (d loss / d logits written to a second buffer)
    K2+K3  ONE loss + 32-stat launch over all tokens of the step
    N > 1: one all-gather of the stats vector across ranks AND the data-parallel gradient all-reduce of a
    7B learner (15.2 GB of bf16 gradients in 1 GiB buckets over RCCL) - without it the sharded path has no
    exchange step and a scaling figure says nothing about a DP learner

`value` is timed with EVERY logits row read (the reference's finiteness assert covers all positions, rl/__init__.py:213); the
opt-out that leaves unlabelled rows unread is timed on one step afterwards (`value_skip_unlabelled`).

The transformer forward/backward itself is outside the path (stock PyTorch-ROCm); the metric string says so, and
`--detail` measures a model-in-the-loop step separately.  Scaling is STRONG: the global batch of 4096 sequences is fixed and
sharded across ranks.  Rank 0 prints ONE JSON line of < 4 KB (`split_line`: the driver's keys + roofline + cpu_baseline +
weight_sync) and writes everything else it measured to `--detail-out` (default gpurun_out/bench_detail.json).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

T_START = time.perf_counter()
LINE_LIMIT_BYTES = 4096  # the driver keeps the tail of stdout: a line that does not fit in it is an unparsed (= unmeasured) round
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, same guide (AMD's 5 PF figure includes 2:1 sparsity)

WORKLOADS = {
    # name: (global batch, seq_length, vocab, attempts)
    "7b_grpo_bs4096_seq8192": (4096, 8192, 152064, 8),        # BASELINE.json configs[2] / [3] (the headline)
    "0p5b_grpo_bs512_seq2048": (512, 2048, 151936, 8),        # configs[1]
    "32b_grpo_kl_bs4096_seq8192": (4096, 8192, 152064, 8),    # configs[4]: KL-to-reference on, TP = 2 receivers
    "tiny": (64, 512, 4096, 8),
}
# name: (parameter set of weight_sync_probe.qwen25_shapes, hidden size of the output head, kl_coef, bf16 gradient bytes of a DP learner)
WORKLOAD_MODEL = {
    "7b_grpo_bs4096_seq8192": ("7b", 3584, 0.0, 15_231_233_024),
    "0p5b_grpo_bs512_seq2048": ("0p5b", 896, 0.0, 988_065_536),
    "32b_grpo_kl_bs4096_seq8192": ("32b", 5120, 0.001, 65_527_752_704),  # kl_coef: conf/deepscaler15b.yaml:34, SURVEY §8(d)
    "tiny": ("0p5b", 256, 0.0, 1 << 20),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default=os.environ.get("PRL_BENCH_WORKLOAD", "7b_grpo_bs4096_seq8192"), choices=list(WORKLOADS))
    p.add_argument("--logits-mode", default=os.environ.get("PRL_BENCH_LOGITS_MODE", "fused"), choices=["fused", "two_pass"])
    p.add_argument("--detail", action="store_true",
                   help="also run the side measurements (MFMA head roofline, model-in-the-loop step, reference-policy head, preprocessor loop, "
                        "configs[1] pipeline, transport) - they go to the detail file, never into the printed line")
    p.add_argument("--detail-out", default=os.environ.get("PRL_BENCH_DETAIL_OUT", "gpurun_out/bench_detail.json"),
                   help="where rank 0 writes everything that is not on the printed line (relative to the repo root)")
    p.add_argument("--budget-s", type=float, default=float(os.environ.get("PRL_BENCH_BUDGET_S", 270)),
                   help="wall-clock budget of a default run: the optional leg (live PMC passes) is skipped, and says so, when it would not fit")
    p.add_argument("--skip-unlabelled-steps", type=int, default=int(os.environ.get("PRL_BENCH_OPTOUT_STEPS", 1)),
                   help="steps timed AFTER the timed region with HotPathStep(skip_unlabelled=True) for `value_skip_unlabelled` (0 = leave it out)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-weight-sync", action="store_true")
    p.add_argument("--no-fused-head", action="store_true", help="--detail: skip the MFMA fused-head measurement (roofline_mfma)")
    p.add_argument("--no-grad-allreduce", action="store_true", help="N > 1: leave the gradient-sized all-reduce out of the step")
    p.add_argument("--grad-bytes", type=int, default=int(os.environ.get("PRL_BENCH_GRAD_BYTES", 0)),
                   help="N > 1: bytes of data-parallel gradients all-reduced per step (default: the workload's model in bf16)")
    p.add_argument("--no-e2e", action="store_true", help="--detail: quote the committed model-in-the-loop step from profiles/ instead of running it (source: committed)")
    p.add_argument("--no-live-pmc", action="store_true", help="quote the committed PMC traffic figure instead of measuring it with two rocprofv3 --pmc sub-runs")
    p.add_argument("--no-transport", action="store_true", help="--detail: skip the host-side transport probe (shm log / files backend round trips)")
    p.add_argument("--no-preprocess-loop", action="store_true", help="--detail: skip the actor-record -> published micro-batch measurement (preprocess_loop)")
    p.add_argument("--no-ref-logprob", action="store_true", help="--detail: skip the reference-policy head measurement (ref_logprob)")
    p.add_argument("--no-pipeline", action="store_true",
                   help="--detail: skip BASELINE configs[1] run AS a pipeline (actor -> preprocessor -> learner -> engine, four processes on this GPU)")
    p.add_argument("--pipeline-steps", type=int, default=int(os.environ.get("PRL_BENCH_PIPELINE_STEPS", 4)), help="optimizer steps of the pipeline run (the first is warm-up)")
    p.add_argument("--cpu-baseline-threads", default=None,
                   help="comma-separated thread counts: time ONLY the cpu_baseline loss leg at each count and exit (no GPU work)")
    p.add_argument("--backend", default=os.environ.get("PRL_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                   help="gloo + PRL_BENCH_SHARE_DEVICE=1 runs N ranks on ONE GPU (dry run of the N > 1 logic)")
    return p.parse_args()


class EventTimer:
    """HIP events on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.pairs: dict[str, list] = {}

    def time(self, name):
        timer = self

        class _Ctx:
            def __enter__(self_inner):
                self_inner.a = torch.cuda.Event(enable_timing=True)
                self_inner.b = torch.cuda.Event(enable_timing=True)
                self_inner.a.record()

            def __exit__(self_inner, *exc):
                self_inner.b.record()
                timer.pairs.setdefault(name, []).append((self_inner.a, self_inner.b))

        return _Ctx()

    def summary(self) -> dict[str, dict]:
        out = {}
        for name, pairs in self.pairs.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = {"launches": len(ms), "avg_us": 1e3 * float(np.mean(ms)), "min_us": 1e3 * float(np.min(ms))}
        return out


def usable_host_cores() -> int:
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU
    boxes show 256 CPUs but grant 16 - oversubscribing the quota makes the torch leg 3x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_thread_sweep(counts: list[int], seq_length: int, vocab: int, t_logits: int = 1024) -> None:
    """Thread sweep of the cpu_baseline loss leg (profiles/r01p_cpu_baseline_threads.txt): which
    thread count is fair to report on this host."""
    from oracle import preprocess as opre
    from oracle import rl_loss_torch as orlt
    from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries

    print("nproc", os.cpu_count(), "usable", usable_host_cores(), flush=True)
    t_logits = min(t_logits, seq_length)
    rag, reasons = make_ragged(1, attempts=8, seq_length=seq_length, vocab=vocab, seed=99, dense=True)
    batch = opre.collate_packed([opre.preprocess_chunk(ragged_to_entries(rag, reasons), 2, False)[0]], 2, 1)
    b = {k: (v[:, :t_logits] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in batch.items()}
    cfg = dict(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
               clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, batch_size=4096, temperature=1.0)
    logits = (np.random.default_rng(0).standard_normal((1, t_logits, vocab)) * 2).astype(np.float32)
    for n in counts:
        torch.set_num_threads(n)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            orlt.rl_step_closed_form(logits, b, cfg, 0, 10, True)
            ts.append(time.perf_counter() - t0)
        print(f"threads {n:4d}: {min(ts[1:]) / t_logits * 1e6:8.1f} us/token", flush=True)


def cpu_baseline(seq_length: int, vocab: int) -> dict:
    """The CPU oracle (numpy / torch-CPU restatement of the reference, pinned to it by golden vectors) timed on this box's
    host cores, leg by leg as BASELINE.md §2 lists them, on a bounded sample of the same workload; beside every leg the
    REFERENCE's own function timed in the build container (profiles/r03_reference_cpu_legs.json - the GPU box has no
    /root/reference).  A reported baseline only."""
    import statistics
    import tempfile

    from oracle import preprocess as opre
    from oracle import rl_loss as orl
    from oracle import rl_loss_torch as orlt
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries

    n_seq, t_logits = 64, min(seq_length, 2048)
    rag, reasons = make_ragged(n_seq // 8, attempts=8, seq_length=seq_length, vocab=vocab, seed=99, dense=True)
    entries = ragged_to_entries(rag, reasons)
    n_tok = sum(len(e["input_ids"]) for e in entries)
    cfg = dict(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
               clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, batch_size=4096, temperature=1.0)

    def median_of(fn, reps, warm, budget_s=12.0):
        """Median of `reps` timed calls after `warm` untimed ones; the leg is BOUNDED: once `budget_s` is spent no further
        warm-up or repetition is started (at least one call is always timed), so a host with slow page faults cannot turn
        the baseline into minutes."""
        spent = time.perf_counter()
        for _ in range(warm):
            if time.perf_counter() - spent > budget_s / 2:
                break
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - spent > budget_s:
                break
        return statistics.median(ts)

    legs: dict[str, dict] = {}
    # -- leg 1: preprocess_fn + populate_rl_data; leg 2: collate_packed (single process, python lists like the reference)
    t_prep = median_of(lambda: opre.preprocess_chunk(entries, 2, False), 3, 1)
    data = opre.preprocess_chunk(entries, 2, False)
    t_coll = median_of(lambda: [opre.collate_packed([d], 2, 1) for d in data], 3, 1)
    batches = [opre.collate_packed([d], 2, 1) for d in data]
    legs["preprocess"] = {"us_per_token": 1e6 * t_prep / n_tok, "samples_per_s": n_seq / t_prep}
    legs["collate_packed"] = {"us_per_token": 1e6 * t_coll / n_tok}
    t_pre = (t_prep + t_coll) / n_seq  # s per sequence
    # -- leg 3: the files-backend wire (the reference's JSONL record format) round trip of 16 micro-batches
    wire_n = 16
    pbs = [PipelineBatchEncoding(**{k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in b.items()})
           for b in batches[:wire_n]]
    wire_tok = sum(int(b.input_ids.numel()) for b in pbs)
    with tempfile.TemporaryDirectory() as tmp:
        def wire():
            w = streams.FileStreamWriter(streams.SingleStreamSpec(exp_path=Path(tmp), topic="w", partition=0), "w")
            with w:
                for b in pbs:
                    w.write(b)
            with streams.FileStreamReader(streams.SingleStreamSpec(exp_path=Path(tmp), topic="w", partition=0)) as r:
                for k, rec in enumerate(r.read()):
                    PipelineBatchEncoding(**rec)
                    if k + 1 == wire_n:
                        break

        t_wire = median_of(wire, 2, 1)
        wire_bytes = (Path(tmp) / "streams" / "w" / "0" / "0" / "0.jsonl").stat().st_size
    legs["wire"] = {"us_per_token": 1e6 * t_wire / wire_tok, "bytes_per_token": wire_bytes / wire_tok, "micro_batches": wire_n}
    # -- legs 4-6: the post-model loss path on a slice of one micro-batch [t, V]
    b = {k: (v[:, :t_logits] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in batches[0].items()}
    rng = np.random.default_rng(0)
    cores = usable_host_cores()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        # V = 8: token loss + reduce + statistics alone (K2 + K3)
        b8 = dict(b)
        b8["input_ids"] = b["input_ids"] % 8
        lg8 = (rng.standard_normal((1, t_logits, 8)) * 2).astype(np.float32)
        nlp8, ent8 = orl.logprob_entropy(lg8, b8["input_ids"], 1.0)[:2]
        t_k23 = median_of(lambda: orl.token_loss(b8, nlp8, ent8, cfg, 0, 10, True), 5, 2)
        legs["loss_v8"] = {"us_per_token": 1e6 * t_k23 / t_logits, "tokens": t_logits}
        # V = vocab: (1) forward only, (2) forward + closed-form gradient with vectorised fp32 torch CPU kernels over every host
        # core (torch's CPU logsumexp backward, which the reference's autograd uses, is a scalar loop an order of magnitude slower)
        logits = (rng.standard_normal((1, t_logits, vocab)) * 2).astype(np.float32)
        t_both = median_of(lambda: orlt.rl_step_closed_form(logits, b, cfg, 0, 10, True), 3, 1)
        t_fwd = median_of(lambda: orlt.rl_step(logits, b, cfg, 0, 10, True, want_grad=False), 3, 1, budget_s=8.0)
        legs["logprob_fwd"] = {"us_per_token": 1e6 * t_fwd / t_logits, "tokens": t_logits, "vocab": vocab}
        legs["logprob_fwd_bwd_closed_form"] = {"us_per_token": 1e6 * t_both / t_logits, "tokens": t_logits, "vocab": vocab}
    finally:
        torch.set_num_threads(prev_threads)
    t_loss_tok = t_both / t_logits  # s per token
    t_np = int(os.environ.get("PRL_BENCH_CPU_NUMPY_TOKENS", 512))
    t0 = time.perf_counter()
    orl.rl_step(logits[:, :t_np], {k: (v[:, :t_np] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in b.items()}, cfg, 0, 10, True)
    t_np_tok = (time.perf_counter() - t0) / t_np
    per_sample = t_pre + t_loss_tok * seq_length
    ref = _committed_json("profiles/r03_reference_cpu_legs.json",
                          note="the REFERENCE's own functions (preprocess_fn + populate_rl_data, collate_packed, JSONL record format, rl_step V = 8, "
                               "rl_step V = 152 064 forward / autograd backward) timed ONCE in the build container on its 8 cores (the GPU box has no "
                               "/root/reference); stated constants, not re-measured here")
    if ref and "legs" in ref:
        for name, leg in legs.items():
            r = ref["legs"].get(name)
            if r:
                leg["reference_us_per_token"] = r["us_per_token"]
        if "logprob_fwd" in ref["legs"] and "logprob_bwd" in ref["legs"]:
            legs["logprob_fwd_bwd_closed_form"]["reference_us_per_token"] = ref["legs"]["logprob_fwd"]["us_per_token"] + ref["legs"]["logprob_bwd"]["us_per_token"]
            legs["logprob_fwd_bwd_closed_form"]["reference_note"] = "reference = forward + AUTOGRAD backward"
    port = {
        "value": 1.0 / per_sample,
        "unit": "samples/s",
        "cores": cores,
        "kind": "port",
        "measured_in_this_run": True,
        "sample": f"oracle (golden-pinned port of the reference): preprocess+collate of {n_seq} x {seq_length}-token sequences, one process "
                  f"({t_pre * 1e3:.1f} ms/seq) + logits->loss->dlogits on {t_logits} tokens x V={vocab}, vectorised fp32 torch CPU kernels "
                  f"(closed-form gradient), {cores} threads, median of 3 ({t_loss_tok * 1e6:.0f} us/token); extrapolated to {seq_length}-token samples",
        "legs": legs,
        "legs_note": "BASELINE.md §2 legs: the port measured on THIS box (us_per_token) next to the reference's own function measured in the build "
                     "container (reference_us_per_token, 8 cores of a different host)",
        "scalar_port": {"value": 1.0 / (t_pre + t_np_tok * seq_length), "cores": 1,
                        "sample": f"same path, single-thread numpy on {t_np} tokens ({t_np_tok * 1e6:.0f} us/token)"},
        "host": {"nproc": os.cpu_count(), "cgroup_cpu_quota": cores},
    }
    port["reference"] = ref
    return port


def transport_probe(seq_length: int, vocab: int) -> dict:
    """Throughput of the transport rows: 64 micro-batches of `seq_length` tokens as
    `PipelineBatchEncoding` records through the shm log (binary SoA, futex-parked reader) and 16 of them through the files
    backend (the reference's JSONL format), write and read + decode timed separately on the host; one group of 8 rollouts
    as a `PRLROL01` record next to its JSONL text record.  Host-side; extra object, never `value`."""
    import tempfile

    from pipelinerl_amd import batch_codec, streams
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries

    T = seq_length
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, vocab, (1, T), generator=g)
    f = lambda: torch.randn(1, T, generator=g)  # noqa: E731
    pb = PipelineBatchEncoding(input_ids=ids, labels=ids.clone(), attention_mask=torch.ones_like(ids), position_ids=torch.arange(T)[None],
                               segment_ids=torch.zeros_like(ids), rewards=f(), advantages=f(), ref_logprobs=f(), old_logprobs=f(),
                               group_tokens=f().abs() + 1, num_labels=f().abs() + 1, overflow=torch.zeros(1, T), model_version=0, is_packed=True,
                               seq_boundaries=torch.tensor([0, T], dtype=torch.int32))
    out: dict = {"record": f"PipelineBatchEncoding [1, {T}] (5 x int64 + 7 x fp32 columns)"}
    with tempfile.TemporaryDirectory() as tmp:
        was = (streams._backend, dict(streams._backend_options))
        try:
            for backend, n in (("shm", 64), ("files", 16)):
                streams.reset_streams_backend()
                streams.set_streams_backend(backend, **({"segment_bytes": 64 << 20, "trim_topics": ()} if backend == "shm" else {}))
                spec = streams.SingleStreamSpec(exp_path=Path(tmp), topic=f"bench_{backend}", partition=0)
                t0 = time.perf_counter()
                with streams.write_to_streams(spec) as w:
                    for _ in range(n):
                        w.write(pb)
                t1 = time.perf_counter()
                with streams.read_stream(spec) as r:
                    for k, rec in enumerate(r.read()):
                        PipelineBatchEncoding(**rec)
                        if k + 1 == n:
                            break
                t2 = time.perf_counter()
                out[f"{backend}_write_us_per_token"] = 1e6 * (t1 - t0) / (n * T)
                out[f"{backend}_read_us_per_token"] = 1e6 * (t2 - t1) / (n * T)
                out[f"{backend}_us_per_token"] = 1e6 * (t2 - t0) / (n * T)
                if backend == "files":
                    out["files_bytes_per_token"] = (Path(tmp) / "streams" / spec.topic / "0" / "0" / "0.jsonl").stat().st_size / (n * T)
            out["shm_bytes_per_token"] = len(batch_codec.encode_batch(pb)) / T
            streams.clean_shm_streams(tmp)
        finally:
            streams.reset_streams_backend()
            if was[0] is not None:
                streams.set_streams_backend(was[0], **was[1])
    rag, reasons = make_ragged(1, attempts=8, seq_length=seq_length, vocab=vocab, seed=5, dense=True)
    n_tok = int(rag.host_seq_off[-1])
    t0 = time.perf_counter()
    rec = batch_codec.encode_rollouts(rag)
    batch_codec.decode(rec)
    t1 = time.perf_counter()
    text = json.dumps(ragged_to_entries(rag, reasons))
    out["rollout_record"] = {"PRLROL01_bytes_per_token": len(rec) / n_tok, "PRLROL01_codec_us_per_token": 1e6 * (t1 - t0) / n_tok,
                             "jsonl_bytes_per_token": len(text) / n_tok, "what": "one group of 8 x %d-token rollouts (the `actor` stream record)" % seq_length}
    out["reference_wire_us_per_token"] = (_committed_json("profiles/r03_reference_cpu_legs.json", note="") or {}).get("legs", {}).get("wire", {}).get("us_per_token")
    return out


def _committed_json(rel: str, note: str) -> dict | None:
    """A measurement committed under profiles/ by an earlier, separate run: quoted with its source."""
    f = ROOT / rel
    if not f.exists():
        return None
    try:
        d = json.loads(f.read_text().splitlines()[0])
    except Exception:  # noqa: BLE001
        return None
    d["source"] = rel
    d["note"] = note
    return d


def live_pmc_traffic(kernel_substr: str = "fused_logits_loss_keep_kernel") -> dict | None:
    """HBM-side bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two separate `rocprofv3 --pmc` passes
    (FETCH_SIZE, then WRITE_SIZE; kernel trace only, as the CDNA guide prescribes) over `scripts/kernel_sweep.py --quick` (the
    same kernel on the same [8192, 152 064] fp32 micro-batch), in subprocesses.  FETCH_SIZE is doubled (gfx950 tallies the 128-byte
    requests of a 16 B/lane stream at 64 bytes; calibrated on a 1 GiB copy in round 1), WRITE_SIZE taken as is, both in KB.
    Returns None when rocprofv3 is unavailable or a pass fails (the committed figure is quoted instead, labelled)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        return None
    # this process is itself being profiled (rocprofv3 -- python bench.py): no profiler inside a profiler
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = Path(tmp) / counter
            try:
                r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", str(out), "-o", "pmc", "--",
                                    sys.executable, str(ROOT / "scripts" / "kernel_sweep.py"), "--quick"],
                                   cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True,
                                   timeout=float(os.environ.get("PRL_BENCH_PMC_TIMEOUT", 90)))
            except Exception:  # noqa: BLE001
                return None
            files = glob.glob(str(out / "**" / "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            per_dispatch: dict[str, float] = {}
            with open(files[0], newline="") as fh:
                for row in csv.DictReader(fh):
                    if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        per_dispatch[row["Dispatch_Id"]] = per_dispatch.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per_dispatch:
                return None
            got[counter] = (sum(per_dispatch.values()) / len(per_dispatch), len(per_dispatch))
    fetch_kb, n = got["FETCH_SIZE"]
    write_kb, _ = got["WRITE_SIZE"]
    return {"hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0, "fetch_kb": fetch_kb, "write_kb": write_kb, "launches": n}


def fused_head_probe(dev: torch.device, seq_length: int, vocab: int, hidden: int) -> dict:
    """The MFMA-bound part of the path: hidden states -> new_logprobs / entropy and back, logits
    never written.  One micro-batch of `seq_length` tokens; HIP events on torch's current stream (where the
    kernels are launched).  flops counted as executed bf16 MFMA work: every plane product of the 2-term split."""
    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead

    T, H, V = seq_length, hidden, vocab
    g = torch.Generator(device=dev).manual_seed(5)
    h = torch.empty(1, T, H, device=dev).normal_(generator=g).to(torch.bfloat16)
    W = torch.empty(V, H, device=dev).normal_(0.0, 0.02, generator=g)
    ids = torch.randint(3, V, (1, T), device=dev, generator=g)
    labels = ids.clone()
    labels[:, : T // 16] = -100
    old = -torch.empty(1, T, device=dev).normal_(generator=g).abs() * 0.7
    z = torch.zeros(1, T, device=dev)
    batch = PipelineBatchEncoding(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), position_ids=torch.arange(T, device=dev)[None],
                                  old_logprobs=old, ref_logprobs=old.clone(), advantages=torch.empty(1, T, device=dev).normal_(generator=g), rewards=z.clone(),
                                  group_tokens=torch.full((1, T), 5000.0, device=dev), num_labels=torch.full((1, T), float(T - T // 16), device=dev),
                                  overflow=z.clone(), model_version=0, is_packed=True)
    cfg, _, _ = make_loss_config(RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, batch_size=4096,
                                          clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False), 0, 10)
    head = FusedLmHead(W)
    head.refresh()

    def timed(fn, iters):
        fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))

    fwd_ms = timed(lambda: head.logprob_entropy(h, ids, 1.0), 5)
    keep_ms = timed(lambda: head.logprob_entropy(h, ids, 1.0, keep=True), 5)  # the training forward: also leaves the logits behind
    nlp, ent, lse2, hb, kept = head.logprob_entropy(h, ids, 1.0, keep=True)
    _, _, g_nlp, _ = grpo_loss_from_logprobs(cfg, batch, nlp, ent)
    gw = torch.zeros(V, H, device=dev)
    bwd_ms = timed(lambda: head.backward_from_token_grads(hb, ids, 1.0, lse2, ent, g_nlp, None, None, grad_weight=gw, kept_logits=kept), 2)
    del kept
    torch.cuda.empty_cache()
    rec_ms = timed(lambda: head.backward_from_token_grads(hb, ids, 1.0, lse2, ent, g_nlp, None, None, grad_weight=gw), 2)
    gemm = 2.0 * T * V * H
    fwd_tf = 2 * gemm / (fwd_ms * 1e-3) / 1e12
    chunk_rows = head.chunk_rows
    del gw, head
    torch.cuda.empty_cache()
    del W
    torch.cuda.empty_cache()
    return {
        "bound": "mfma", "kernel": "lmhead_fwd_kernel<CfgDual: 256x256x32 dual-plane tile, ring of 3 LDS stages, phase-shifted hand-placed stream, v_mfma_f32_32x32x16_bf16>", "achieved": fwd_tf, "peak": MFMA_BF16_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": fwd_tf / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
        "flops_per_launch": 2 * gemm, "ms_per_launch": fwd_ms,
        "config": {"tokens": T, "hidden": H, "vocab": V, "weight": "fp32 as two bf16 planes (fp32-GEMM accuracy)", "logits_materialised_bytes": 0},
        "forward_keeping_logits": {"ms": keep_ms, "executed_tflops": 2 * gemm / (keep_ms * 1e-3) / 1e12, "kept_bytes": 4 * T * V,
                                   "what": "the same launch, each accumulator tile also stored as fp32 logits (16-byte stores from the accumulator "
                                           "layout): the forward of a training step (FusedLmHead(keep_logits=True), the default)"},
        "backward": {"ms": bwd_ms, "executed_tflops": 5 * gemm / (bwd_ms * 1e-3) / 1e12, "chunk_rows": chunk_rows,
                     "what": "from the kept logits: one elementwise pass (fp32 logits -> d logits as two ROW-MAJOR bf16 planes) + d hidden (3 products on the "
                             "triple-plane core, one contraction slice per XCD) + d W (2 products, fragments gathered from the row-major planes by "
                             "ds_read_b64_tr_b16)"},
        "backward_recompute": {"ms": rec_ms, "executed_tflops": 7 * gemm / (rec_ms * 1e-3) / 1e12,
                               "what": "FusedLmHead(keep_logits=False): no logits anywhere, the d-logits planes come from recomputing both plane products "
                                       "(a workgroup walks a range of vocabulary tiles) - 7 products instead of 5"},
        "fp32_equivalent_tflops": gemm / (fwd_ms * 1e-3) / 1e12,
        "note": "second roofline object for the MFMA-bound fused output head (hidden -> log-prob/entropy, logits never in HBM); "
                "`roofline` above stays the HBM-bound kernel that dominates `value`",
    }


def preprocess_loop_probe(dev: torch.device, seq_length: int, vocab: int, attempts: int) -> dict:
    """The preprocessor LOOP at the reference's granularity (preprocess.py:370-704, `chunk_n_groups` = 2): groups of
    `attempts` x `seq_length`-token rollouts are written to the `actor` stream first - as PRLROL01 records on the shm log
    and as the reference's JSONL list-of-dicts records on the files backend - then `PreprocessorLoop.run` consumes them:
    reader thread -> decode -> one H2D per chunk -> K5 -> scheduler -> K6 per drain -> one D2H -> encode -> per-trainer
    `training_data` partitions.  Timed: `run()` from its first line to the last published micro-batch (wall clock; stream
    read + decode run concurrently in the loader thread, as in the reference).  Host phases come from `perf_counter`
    pairs inside the loop, K5 / K6 device time from HIP event pairs on the loop's stream.  Extra object, never `value`."""
    import shutil
    import tempfile

    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries

    shm_free = shutil.disk_usage("/dev/shm").free if Path("/dev/shm").exists() else 0
    tok_per_group = attempts * seq_length
    # shm: the actor records (12 B/token) and the published batches (68 B/token) of one case live in /dev/shm together
    # 128 groups = 64 chunks per case: the loop's fill and drain (first chunk's latency, the last drain's publish) are ~ one chunk
    # each, which was ~10 % of the 16-chunk runs of rounds 4 and 5 (r05j and before) and is 2-3 % here
    n_fast = int(max(4, min(int(os.environ.get("PRL_BENCH_PREPROCESS_GROUPS", 128)), shm_free // 4 // (tok_per_group * 80)))) // 2 * 2
    n_text = 4
    rl = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, clamp_log_ratio_ref_new_value=5,
                  temperature=1.0, divide_advantage_by_std=False, group_normalization=False, batch_size=4096)
    groups = [make_ragged(1, attempts=attempts, seq_length=seq_length, vocab=vocab, seed=4000 + g, dense=True) for g in range(n_fast)]
    was = (streams._backend, dict(streams._backend_options))
    out: dict = {"what": f"actor stream -> PreprocessorLoop (chunk_n_groups = 2, packed, seq_length {seq_length}) -> training_data; groups of "
                         f"{attempts} x {seq_length}-token rollouts; wall clock of run() incl. stream read + decode (loader thread)", "cases": {}}

    def one_case(backend: str, binary: bool, trainers: int, n_groups: int, batched: bool, overlap: bool = True, wire: str = "full", consumer: bool = False):
        tmp = tempfile.mkdtemp(prefix="prl_bench_pre_")
        try:
            streams.reset_streams_backend()
            streams.set_streams_backend(backend, **({"segment_bytes": 256 << 20, "trim_topics": (), "owner": True} if backend == "shm" else {}))
            spec = streams.SingleStreamSpec(exp_path=Path(tmp), topic="actor")
            with streams.write_to_streams(spec) as w:
                for rag, reasons in groups[:n_groups]:
                    w.write(rag if binary else ragged_to_entries(rag, reasons))
            cfg = PreprocessorConfig(exp_path=Path(tmp), num_trainers=trainers, train_batch_size=1, gradient_accumulation_passes=4096,
                                     seq_length=seq_length, attempts=attempts, rl=rl, eos_token_id=2, chunk_n_groups=2,
                                     pop_old_data=False)  # lossless: with the default the loader DROPS old chunks when the loop is the slower side
            loop = PreprocessorLoop(cfg, dev, batched_transfers=batched, profile=True, overlap_publish=overlap, wire=wire)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # the packer flushes a micro-batch when the NEXT sample no longer fits (preprocess.py:610-625): the very last
            # sample of the stream stays pending, so the target is one short of what was written
            target = n_groups * attempts - 1
            n = loop.run(max_published_samples=target, idle_timeout=5.0)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tokens = n * seq_length  # dense rollouts: every sample is seq_length tokens
            chunks = n_groups // 2
            prof = dict(loop.prof)
            kern = loop.kernel_seconds()
            planning = sum(prof.get(k, 0.0) for k in ("ingest_flatten", "k5_plan", "k5_launch", "schedule", "k6_plan_launch"))
            res = {"published_samples": n, "tokens": tokens, "chunks": chunks, "wall_s": dt, "tokens_per_s": tokens / dt, "us_per_token": 1e6 * dt / tokens,
                   "host_phase_us_per_chunk": {k: 1e6 * v / chunks for k, v in sorted(prof.items())},
                   "kernel_us_per_chunk": {k: 1e6 * v / chunks for k, v in kern.items()},
                   "host_planning_frac": planning / dt,
                   "host_planning_is": "ingest_flatten + k5_plan + k5_launch + schedule + k6_plan_launch (plan AND launch calls; transfers, codec, publish, waiting for input excluded)"}
            if loop.stager is not None:
                res["transfers_per_chunk"] = {"h2d": loop.stager.uploads / chunks, "d2h": loop.down_stager.downloads / chunks,
                                              "h2d_bytes": loop.stager.bytes_up / chunks, "d2h_bytes": loop.down_stager.bytes_down / chunks}
            if loop.publisher_ns[0]:
                res["publish"] = ("native publisher thread (csrc/prl_publish.cpp): device -> host copy + gathering of drain k's records into the logs overlap ingest / "
                                  "K5 / K6 of drain k + 1; the loop itself only builds headers and a piece table (publish_submit)")
                res["publisher_us_per_chunk"] = {"busy": 1e-3 * loop.publisher_ns[0] / chunks, "d2h": 1e-3 * loop.publisher_ns[1] / chunks}
            else:
                res["publish"] = "inline (device -> host copy, framing and append inside the loop: d2h + encode_publish)"
            if wire == "compact":
                res["publish"] = ("compact wire: no K6 and no per-token device -> host traffic here; the native publisher gathers every micro-batch's ragged "
                                  "columns from the decoded actor records on the host (PRLCMP01, 12-16 B/token); K6 runs in the learner's loader")
            assert n == target, f"published {n} of {target} samples"
            if consumer:  # the other end of the wire: what the learner's loader thread does per record (finetune_loop.RecordToBatch)
                from pipelinerl_amd.finetune_loop import RecordToBatch
                from pipelinerl_amd.ring import Log

                to_batch = RecordToBatch(dev, annotate=True)
                part = streams.SingleStreamSpec(exp_path=Path(tmp), topic="training_data", partition=0)
                want = Log(streams.ring_name(part), reader=True).stats()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                got = toks = 0
                with streams.read_stream(part) as r:
                    for rec in r.read():
                        b = to_batch(rec)
                        toks += int(b.input_ids.shape[1]) if not b.sentinel else 0
                        got += 1
                        if got == want["records"]:
                            break
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                res["consumer"] = {"what": "learner's loader over partition 0: log read + decode + host facts + upload" + (" + K6 on the learner's GPU" if wire == "compact" else " (12 columns)"),
                                   "records": got, "tokens": toks, "log_bytes_per_token": want["bytes"] / max(toks, 1), "us_per_token": 1e6 * dt1 / max(toks, 1)}
            return res
        finally:
            if backend == "shm":
                streams.clean_shm_streams(tmp)
            shutil.rmtree(tmp, ignore_errors=True)

    try:
        one_case("shm", True, 1, min(8, n_fast), True)  # warm-up: library, page-locked ring, the allocator's block cache (several chunks are alive at once)
        out["cases"]["PRLROL01_to_shm_1_trainer"] = one_case("shm", True, 1, n_fast, True, consumer=True)
        out["cases"]["PRLROL01_to_shm_4_trainers"] = one_case("shm", True, 4, n_fast, True)
        one_case("shm", True, 1, min(8, n_fast), True, wire="compact", consumer=True)  # warm-up of the compact path (publisher without a block, the loader's ring)
        out["cases"]["PRLROL01_to_shm_1_trainer_compact_wire"] = one_case("shm", True, 1, n_fast, True, wire="compact", consumer=True)
        out["cases"]["PRLROL01_to_shm_4_trainers_compact_wire"] = one_case("shm", True, 4, n_fast, True, wire="compact")
        out["cases"]["PRLROL01_to_shm_1_trainer_inline_publish"] = one_case("shm", True, 1, n_fast, True, overlap=False)  # the round-4 loop
        out["cases"]["PRLROL01_to_shm_1_trainer_one_copy_per_array"] = one_case("shm", True, 1, n_fast, False)
        out["cases"]["JSONL_to_files_1_trainer"] = one_case("files", False, 1, n_text, True)
    finally:
        streams.reset_streams_backend()
        if was[0] is not None:
            streams.set_streams_backend(was[0], **was[1])
    ref = (_committed_json("profiles/r03_reference_cpu_legs.json", note="") or {}).get("legs", {})
    if ref:
        r_pre, r_col, r_wire = (ref.get(k, {}).get("us_per_token") for k in ("preprocess", "collate_packed", "wire"))
        out["reference_us_per_token"] = {"preprocess_fn+populate_rl_data": r_pre, "collate_packed": r_col, "jsonl_wire_round_trip": r_wire,
                                         "source": "profiles/r03_reference_cpu_legs.json (the reference's own functions, 8 cores of the build container)"}
        best = out["cases"]["PRLROL01_to_shm_1_trainer"]["us_per_token"]
        if r_pre and r_col:
            out["speedup_vs_reference_preprocess_plus_collate"] = (r_pre + r_col) / best
        if r_pre and r_col and r_wire:
            out["speedup_vs_reference_incl_wire"] = (r_pre + r_col + r_wire) / best
    return out


def ref_logprob_probe(dev: torch.device, seq_length: int, vocab: int, hidden: int) -> dict:
    """The reference-policy forward of a KL-enabled config (reference: a second inference server asked over
    HTTP, preprocess.py:86-104, llm.py:606-648): last hidden states [T, H] -> log p_ref of the labelled tokens.
    OLD: stock lm_head GEMM writing `[T, V]` logits + the K1 kernel reading them back (round 3's `annotate_ref_logprobs`).
    FUSED: `FusedLmHead(backward=False)` - the product on the MFMA head, online softmax in its epilogue, no logits.
    Both for an fp32 head (two bf16 planes; the fp32 `F.linear` of finetune/checkpoints.py:87-103 on the old path) and a
    bf16 head (one plane; a bf16 GEMM writing bf16 logits on the old path).  One micro-batch of `seq_length` tokens."""
    from pipelinerl_amd.finetune.rl import logprob_entropy
    from pipelinerl_amd.fused_head import FusedLmHead, token_logprobs_from_hidden

    T, H, V = seq_length, hidden, vocab
    g = torch.Generator(device=dev).manual_seed(11)
    h = torch.empty(1, T, H, device=dev).normal_(generator=g).to(torch.bfloat16)
    ids = torch.randint(3, V, (1, T), device=dev, generator=g)
    labels = ids.clone()
    labels[:, : T // 28] = -100  # the benchmark's prompt share (~3.5 %): below the 1/32 compaction threshold, every row runs

    def timed(fn, iters=5):
        fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))

    out: dict = {"what": f"log p_ref of one {T}-token micro-batch from the reference policy's last hidden states, H = {H}, V = {V}", "heads": {}}
    for name, wdt in (("fp32_head", torch.float32), ("bf16_head", torch.bfloat16)):
        W = torch.empty(V, H, device=dev).normal_(0.0, 0.02, generator=g).to(wdt)
        head = FusedLmHead(W, backward=False, keep_logits=False)
        head.refresh()

        def old_path():
            logits = torch.nn.functional.linear(h.float() if wdt == torch.float32 else h, W)
            nlp = logprob_entropy(logits, ids, 1.0)[0]
            return torch.where(labels != -100, nlp, torch.zeros_like(nlp))

        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        a = old_path()
        old_peak = torch.cuda.max_memory_allocated() - base
        old_ms = timed(old_path)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        b = token_logprobs_from_hidden(head, h, ids, labels, 1.0)
        new_peak = torch.cuda.max_memory_allocated() - base
        new_ms = timed(lambda: token_logprobs_from_hidden(head, h, ids, labels, 1.0))
        planes = 2 if wdt == torch.float32 else 1
        itemsize = 4 if wdt == torch.float32 else 2
        # what each path costs in ACCURACY: the first 256 labelled rows against the fp64 product + fp64 log-softmax
        r0 = T // 28 + 1
        rows = slice(r0, r0 + 256)
        z = h[0, rows].double() @ W.double().t()
        exact = torch.log_softmax(z, -1).gather(1, ids[0, r0 + 1: r0 + 257, None]).squeeze(1)
        err = lambda t: float((t[0, r0 + 1: r0 + 257].double() - exact).abs().max().item())  # noqa: E731
        del z
        entry = {
            "old_ms": old_ms, "fused_ms": new_ms, "speedup": old_ms / new_ms,
            "old_what": ("fp32 F.linear (library GEMM)" if wdt == torch.float32 else "bf16 F.linear (library GEMM)") + f" writing [T, V] {str(wdt).replace('torch.', '')} logits + K1",
            "fused_what": f"{planes} bf16 plane product(s) on v_mfma_f32_32x32x16_bf16, online softmax in the epilogue",
            "fused_executed_tflops": planes * 2.0 * T * V * H / (new_ms * 1e-3) / 1e12,
            "logits_bytes_not_written": T * V * itemsize, "hbm_bytes_saved": 2 * T * V * itemsize,
            "peak_extra_memory_bytes": {"old": old_peak, "fused": new_peak},
            "max_abs_difference": float((a - b).abs().max().item()),
            "max_abs_error_vs_fp64": {"old": err(a), "fused": err(b), "rows": 256},
        }
        if wdt == torch.bfloat16:
            # The bf16 library path above ROUNDS THE LOGITS to bf16 (2^-9 of |logit|, ~1e-2 in a log-prob): it is faster because it is less exact, and
            # outside north_star's 1e-4.  The like-for-like library path keeps the product's fp32 accumulator (bf16 GEMM with an fp32 output, what
            # the reference's fp32 head computes for a bf16 weight, checkpoints.py:87-103) and hands K1 fp32 logits:
            def old_path_fp32_logits():
                logits = torch.mm(h[0], W.t(), out_dtype=torch.float32).unsqueeze(0)
                nlp = logprob_entropy(logits, ids, 1.0)[0]
                return torch.where(labels != -100, nlp, torch.zeros_like(nlp))

            try:
                c = old_path_fp32_logits()
                eq_ms = timed(old_path_fp32_logits)
                entry["old_equal_accuracy"] = {"ms": eq_ms, "speedup": eq_ms / new_ms, "max_abs_error_vs_fp64": err(c),
                                               "what": "bf16 GEMM with fp32 output (torch.mm(out_dtype=float32), library) writing [T, V] fp32 logits + K1: the library "
                                                       "path that meets the same tolerance as the fused head"}
                entry["speedup_at_equal_accuracy"] = eq_ms / new_ms
                del c
            except Exception as e:  # noqa: BLE001 - an older torch has no out_dtype
                entry["old_equal_accuracy"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        out["heads"][name] = entry
        del W, head, a, b
        torch.cuda.empty_cache()
    return out


def weight_sync_probe(rank: int, world: int, dev: torch.device, out: dict) -> dict:
    """Trainer -> actor weight update (rank 0 -> all others) over RCCL, Qwen2.5-7B sized (bf16):
      (1) the wire alone: 15 x 1 GiB buckets, plain broadcast vs scatter + all-gather;
      (2) the whole update as the trainer runs it: flatten 339 parameters into buckets (gather kernel),
          move them (two-stream pipeline), scatter them into the receivers' own weight tensors, verified.
    Results are written into `out` as they arrive (a watchdog may print it if a later stage hangs).
    Extra field, never `value`."""
    import torch.distributed as dist

    total_bytes = int(os.environ.get("PRL_BENCH_WSYNC_BYTES", out.get("param_bytes", 15_231_233_024)))  # default: 7.6B params bf16
    bucket_bytes = 1 << 30
    grp = None
    try:
        from pipelinerl_amd.weight_sync import WeightSyncGroup

        out["stage"] = "init"
        grp = WeightSyncGroup.from_torch_distributed(rank, world, dev)
        n_comm, r_comm = grp.comm_size()  # what RCCL itself says about the communicator, not our launch arguments
        out["rccl_comm_size"], out["rccl_comm_rank"] = n_comm, r_comm
        assert (n_comm, r_comm) == (world, rank), f"RCCL communicator is {r_comm}/{n_comm}, launched as {rank}/{world}"
        bucket = torch.empty(bucket_bytes, dtype=torch.uint8, device=dev)
        n_buckets = (total_bytes + bucket_bytes - 1) // bucket_bytes
        out.update({"bytes": n_buckets * bucket_bytes, "n_receivers": world - 1, "bucket_bytes": bucket_bytes})
        for mode in ("broadcast", "scatter_allgather"):
            out["stage"] = f"wire:{mode}"
            for it in range(2):  # first pass warms the RCCL channels
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_buckets):
                    grp.broadcast_bucket(bucket, mode=mode)
                torch.cuda.synchronize()
                dist.barrier()
                dt = time.perf_counter() - t0
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out[f"{mode}_ms"] = 1e3 * t.item()
            out[f"{mode}_GBps"] = out["bytes"] / t.item() / 1e9
        del bucket
    except Exception as e:  # the probe must never take the benchmark line down
        out["error"] = f"{type(e).__name__}: {e}"
        return out
    try:
        from pipelinerl_amd.weight_sync import BucketedReceiver, BucketedSender, ParamSpec
        from pipelinerl_amd.weight_sync_probe import qwen25_shapes

        out["stage"] = "full_update"
        shapes = qwen25_shapes(out.get("params", "7b"))
        gen = torch.Generator(device=dev).manual_seed(77)  # same values on every rank: receivers can verify
        probe_name = "model.norm.weight"
        if rank == 0:
            params = [(n, torch.empty(s, dtype=torch.bfloat16, device=dev).normal_(generator=gen)) for n, s in shapes]
            sender = BucketedSender(grp, bucket_bytes)
        else:
            expect = None
            dest = {}
            for n, s in shapes:
                t_ = torch.empty(s, dtype=torch.bfloat16, device=dev).normal_(generator=gen)
                if n == probe_name:
                    expect = t_.clone()
                dest[n] = t_.zero_()
            info = [ParamSpec(n, tuple(s), torch.bfloat16) for n, s in shapes]
            receiver = BucketedReceiver(grp, bucket_bytes)
        for it in range(2):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if rank == 0:
                sender.send(params)
            else:
                receiver.receive(info, None, destinations=dest)
            torch.cuda.synchronize()
            dist.barrier()
            dt = time.perf_counter() - t0
        ok = torch.tensor([1 if rank == 0 or torch.equal(dest[probe_name], expect) else 0], device=dev)
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        out["full_update_ms"] = 1e3 * t.item()
        out["full_update_tensors"] = len(shapes)
        out["full_update_verified"] = bool(ok.item())
        # the same keys the 1-GPU (colocated) object carries, so that the two layouts read alike
        nbytes = sum(int(np.prod(s_)) * 2 for _, s_ in shapes)
        out.update({"median_ms": out["full_update_ms"], "tensors": len(shapes), "gbytes": round(nbytes / 1e9, 3),
                    "effective_GBps": round(nbytes / t.item() / 1e9, 1), "layout": f"trainer rank 0 -> {world - 1} receiver GPU(s) over RCCL / xGMI"})
        out["stage"] = "done"
    except Exception as e:  # noqa: BLE001 - keep the wire numbers
        out["full_update_error"] = f"{type(e).__name__}: {e}"
    finally:
        try:
            grp.close()
        except Exception:  # noqa: BLE001
            pass
    return out


def grad_bucket_sizes(total_bytes: int, bucket_bytes: int = 1 << 30) -> list[int]:
    """Byte sizes of the bf16 gradient buckets a data-parallel learner all-reduces per step: `bucket_bytes` each, the remainder last;
    every size even (whole bf16 elements).  A ring all-reduce is bound by the slowest xGMI link at 2 (N - 1) / N of the bytes, so the
    bucket count - not N - sets the number of collectives: 15 for the 7B set, 62 for 32B."""
    if total_bytes < 0 or total_bytes % 2:
        raise ValueError(f"{total_bytes} bytes is not a whole number of bf16 gradients")
    out, left = [], total_bytes
    while left > 0:
        n = min(left, bucket_bytes)
        out.append(n)
        left -= n
    return out


def self_launch(n: int, share_device: bool) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1) and return their exit code.  Refuses when the node has fewer than N devices, unless the dry-run mode
    (PRL_BENCH_SHARE_DEVICE=1, gloo) was asked for."""
    import socket
    import subprocess

    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < n and not share_device:
        print(f"bench.py: --gpus {n} needs {n} HIP devices, {n_dev} visible; not running (a {n}-GPU line measured on fewer devices would be invalid)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(__file__).resolve()), *sys.argv[1:]]
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    return subprocess.call(cmd, env=env)


def pipeline_probe(steps: int, tiny: bool = False) -> dict:
    """BASELINE `configs[1]` (Qwen2.5-0.5B GRPO, one MI355X, actor + learner colocated, synthetic rollouts bs = 512 seq = 2048) run AS the
    pipeline it is (pipelinerl_amd/pipeline_run.py): four OS processes on this GPU - actor harness over the rollout plugin surface, paced by
    max_lag against the propagated weight version; PreprocessorLoop over shm streams; a random-init policy of the 0.5B shape with the fused
    head, AdamW and `StreamedLearnerStep`; the engine-side update manager receiving every optimizer step's weights over HIP IPC.  Reports
    steady-state samples/s (first step excluded), per-stage busy fraction, queue depths, weight-sync request -> ack under load, the lag
    histogram, and the stages' busy seconds per step added up next to the pipelined step time.  Extra object, never `value`."""
    import tempfile

    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    out: dict = {"what": "BASELINE configs[1] as four concurrent processes on ONE GPU: actor (plugin surface, PRLROL01 records, max_lag pacing) -> PreprocessorLoop "
                         "(shm, chunk_n_groups 2) -> StreamedLearnerStep (Qwen2.5-0.5B shape, random init, bf16, tied fused head, AdamW, activations kept) -> "
                         "weight hand-off to the engine-side update manager over HIP IPC after EVERY optimizer step"}
    cases = [("bs512_seq2048", {}, steps), ("bs512_seq2048_pack_budget_8192", {"pack_budget": 8192}, max(2, steps - 1))]
    if tiny:  # the contract test's workload: the same four processes around a two-layer model
        cases = [("bs512_seq2048", {"model": "tiny", "global_batch": 32, "seq_length": 128, "n_problems": 8}, max(2, min(steps, 3)))]
    for name, kw, n_steps in cases:
        if name != "bs512_seq2048" and os.environ.get("PRL_BENCH_PIPELINE_VARIANTS", "1") == "0":
            continue
        exp = tempfile.mkdtemp(prefix="prl_bench_pipeline_")
        try:
            res = run_pipeline(PipelineSpec(exp_path=exp, steps=n_steps, stage_timeout_s=float(os.environ.get("PRL_BENCH_PIPELINE_TIMEOUT", 600)), **kw))
        except Exception as e:  # noqa: BLE001
            res = {"error": f"{type(e).__name__}: {e}"}
        finally:
            import shutil

            shutil.rmtree(exp, ignore_errors=True)
        if "summary" in res:
            entry = dict(res["summary"])
            entry["wall_s_incl_start_up"] = res["wall_s_incl_start_up"]
            entry["stages"] = {k: {kk: v.get(kk) for kk in ("wall_s", "busy_s", "busy_frac", "published_samples", "updates", "micro_batches", "tokens", "init_s")
                                   if kk in v} for k, v in res["stages"].items()}
            entry["per_step"] = res["stages"]["learner"]["per_step"]
        else:
            entry = {"error": res.get("error")}
        if name == "bs512_seq2048":
            out.update(entry)
            out["config"] = ({"model": "Qwen2.5-0.5B shape (494 M parameters, tied head), random init, bf16", "global_batch": 512, "seq_len": 2048, "attempts": 8,
                              "rollouts": "ragged (P ~ U{64..512}, C ~ U{512..2048-P}), SURVEY §8(d)", "pack_budget_tokens": 2048, "max_lag_samples": 512,
                              "weight_update_interval": 1, "optimizer_steps": n_steps} if not tiny else {**kw, "optimizer_steps": n_steps})
        else:
            entry["what"] = ("the same pipeline with `finetune.seq_length` (the packing budget of a micro-batch) at 8192 tokens instead of the longest rollout: "
                             "the 0.5B body is launch-bound at ~1400 tokens per micro-batch (scripts/learner_microbatch_profile.py)")
            out["pack_budget_8192"] = entry
    return out


def e2e_probe(live: bool) -> dict | None:
    """--detail: a MODEL-IN-THE-LOOP step (random-init Qwen2.5-7B shape, stock PyTorch-ROCm forward/backward + AdamW on ONE MI355X, bs 16 x 8192,
    scripts/e2e_learner_bench.py) in a fresh process - or, with --no-e2e, the committed figure labelled as such."""
    import subprocess

    committed = _committed_json("profiles/r02_e2e_learner_7b_fused_head.json",
                                note="measured separately with scripts/e2e_learner_bench.py and committed; `value` is the post-model hot path on resident "
                                     "logits, NOT learner throughput")
    if not live:
        if committed is not None:
            committed["source"] = "committed (profiles/r02_e2e_learner_7b_fused_head.json), NOT measured in this run"
        return committed
    torch.cuda.empty_cache()
    out = ROOT / "gpurun_out" / "bench_e2e_7b.json"
    out.parent.mkdir(exist_ok=True)
    base = [sys.executable, str(ROOT / "scripts" / "e2e_learner_bench.py"), "--model", "7b", "--batch-size", "16", "--seq-len", "8192",
            "--micro-batch", "1", "--fused", "--fused-head", "--steps", "1", "--warmup", "1", "--out", str(out)]

    def run(extra):
        if out.exists():
            out.unlink()
        try:
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=float(os.environ.get("PRL_BENCH_E2E_TIMEOUT", 900)))
            return json.loads(out.read_text().splitlines()[0]) if r.returncode == 0 and out.exists() else {"error": (r.stderr or r.stdout)[-400:]}
        except Exception as e:  # noqa: BLE001 - must never take the benchmark line down
            return {"error": f"{type(e).__name__}: {e}"}

    t0 = time.perf_counter()
    e2e = run([])
    if "error" in e2e:
        if committed is not None:
            committed["source"] = "committed (profiles/r02_e2e_learner_7b_fused_head.json) - the live run failed: " + e2e["error"][-200:]
            return committed
        return e2e
    e2e["source"] = "measured in this run (scripts/e2e_learner_bench.py in a subprocess of bench.py)"
    e2e["wall_s_including_model_init"] = time.perf_counter() - t0
    # the same step with every layer's activations KEPT (no recompute in the backward): what the 288 GB of one MI355X allow
    k = run(["--no-checkpointing"])
    e2e["without_activation_recompute"] = k if "error" in k else {key: k[key] for key in ("s_per_step", "samples_per_s", "tokens_per_s", "peak_memory_GB", "loss")}
    return e2e


def _pick(d: dict | None, keys) -> dict | None:
    return None if d is None else {k: d[k] for k in keys if k in d}


def split_line(full: dict, detail_path: str) -> tuple[dict, dict]:
    """(the ONE printed line, the detail file's content).  The line carries exactly the driver's contract - scalars where the contract
    has scalars, `cpu_baseline.cores` an int, at most LINE_LIMIT_BYTES - and the path of the detail file; everything else that was measured
    (per-kernel table, CPU legs, side measurements) is in the detail file only.  Pure function of `full`: tests/test_bench_line.py."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                       "vs_baseline", "dtype", "data", "config")}
    r = full["roofline"]
    line["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_us", "min_us",
                                               "launches", "algorithmic_bytes_per_launch")}
    if line["roofline"]["traffic_source"]:
        line["roofline"]["traffic_source"] = line["roofline"]["traffic_source"][:200]
    c = full.get("cpu_baseline")
    if c is not None:
        ref = c.get("reference") or {}
        line["cpu_baseline"] = {
            "value": c["value"], "unit": c["unit"], "cores": int(c["cores"]), "kind": c["kind"], "sample": c["sample"][:420],
            # the reference's OWN functions cannot travel to the GPU box (/root/reference is in the build container only): a committed constant from
            # that container's cores, quoted beside the port that was timed here
            "reference_value": ref.get("samples_per_s_extrapolated"), "reference_cores": ref.get("threads"),
            "reference_source": ref.get("source"),
        }
    else:
        line["cpu_baseline"] = None
    w = full.get("weight_sync")
    line["weight_sync"] = _pick(w, ("metric", "transport", "median_ms", "min_ms", "gbytes", "tensors", "effective_GBps", "n_receivers",
                                    "broadcast_ms", "scatter_allgather_ms", "full_update_verified", "error"))
    if line["weight_sync"] and "error" in line["weight_sync"]:
        line["weight_sync"]["error"] = str(line["weight_sync"]["error"])[:200]
    o = full.get("value_skip_unlabelled")
    line["value_skip_unlabelled"] = None if o is None else o["value"]
    line["skip_unlabelled_steps"] = None if o is None else o["steps"]
    # one number per timed kernel: fraction of the 8 TB/s peak on algorithmic bytes (the table behind them is in the detail file)
    line["hbm_frac"] = {k: round(v["hbm_frac"], 4) for k, v in (full.get("kernels") or {}).items() if "hbm_frac" in v}
    ar = (full.get("kernels") or {}).get("grad_allreduce")
    if ar:
        line["grad_allreduce"] = {"avg_ms": ar["avg_us"] / 1e3, "bytes": ar.get("bytes"), "busbw_GBps": ar.get("busbw_GBps")}
    line["loss"] = full.get("loss")
    if full.get("skipped"):
        line["skipped"] = {k: v[:120] for k, v in full["skipped"].items()}
    line["wall_s"] = full.get("wall_s")
    line["detail"] = detail_path
    size = len(json.dumps(line))
    assert size <= LINE_LIMIT_BYTES, f"the bench line is {size} bytes (limit {LINE_LIMIT_BYTES}): move something to the detail file"
    return line, full


def main():
    args = parse_args()
    if args.cpu_baseline_threads:
        _, seq_length, vocab, _ = WORKLOADS[args.workload]
        cpu_baseline_thread_sweep([int(x) for x in args.cpu_baseline_threads.split(",")], seq_length, vocab)
        return
    import torch.distributed as dist

    share_device = os.environ.get("PRL_BENCH_SHARE_DEVICE") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and relays their exit code.  A run that
        # asks for N GPUs NEVER continues as one rank: the line would read "n_gpus": 1 with all of the batch on GPU 0.
        sys.exit(self_launch(args.gpus, share_device))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a line for a job that is not the one asked for")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if torch.cuda.device_count() < world and not share_device:
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible (PRL_BENCH_SHARE_DEVICE=1 + --backend gloo is the dry-run mode)")
    # host-side torch ops (stream decode, codec, the CPU baseline) use the cores this process was GRANTED: the GPU boxes show 256
    # CPUs behind a 16-core quota, and an intra-op pool sized for 256 spends the quota on its own wake-ups
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_host_cores())))
    if share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    if world > 1:
        # preflight: ONE tiny collective under a watchdog, so that a fabric / rendezvous problem of the first multi-GPU run ends with a message
        # that names the rank and the stage instead of a silent hang until the driver's limit (the timed region has no watchdog by design)
        import threading

        ok = threading.Event()
        limit = float(os.environ.get("PRL_BENCH_PREFLIGHT_TIMEOUT", 300))

        def preflight_watchdog():
            if not ok.wait(limit):
                print(f"bench.py: rank {rank}/{world} ({args.backend}, device {local_rank}): the first all-reduce did not complete in {limit:.0f} s - the process "
                      f"group formed but a collective hangs (HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, "
                      f"NCCL_DEBUG={os.environ.get('NCCL_DEBUG')}; rerun with NCCL_DEBUG=INFO)", file=sys.stderr, flush=True)
                os._exit(3)

        threading.Thread(target=preflight_watchdog, daemon=True).start()
        probe = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if args.backend == "nccl":
            torch.cuda.synchronize()
        assert int(probe.item()) == world, f"all-reduce of ones over {world} ranks gave {probe.item()}"
        ok.set()
    # what the process group itself says (never argv): ranks, and how many DISTINCT devices they sit on
    n_ranks, distinct_devices = 1, 1
    if world > 1:
        n_ranks = dist.get_world_size()
        props = torch.cuda.get_device_properties(dev)
        ident = f"{getattr(props, 'uuid', '')}|{getattr(props, 'pci_bus_id', '')}|{getattr(props, 'pci_device_id', '')}|{local_rank if not share_device else 0}"
        idents: list = [None] * n_ranks
        dist.all_gather_object(idents, ident)
        distinct_devices = len(set(idents))
        assert n_ranks == world, f"the process group has {n_ranks} ranks, the launcher announced {world}"

    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.hotpath import HotPathStep
    from pipelinerl_amd.synthetic import make_ragged

    _lib.load()
    bs, seq_length, vocab, attempts = WORKLOADS[args.workload]
    param_set, hidden, kl_coef, grad_bytes_default = WORKLOAD_MODEL[args.workload]
    if not args.grad_bytes:
        args.grad_bytes = grad_bytes_default
    assert bs % (attempts * world) == 0, "global batch must split into whole groups per rank"
    groups_per_rank = bs // attempts // world
    cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=kl_coef, final_kl_coef=kl_coef,
                   clamp_log_ratio_ref_new_value=5, temperature=1.0, divide_advantage_by_std=False,
                   group_normalization=False, batch_size=bs)

    # synthetic rollouts of this rank's shard, resident in HBM before the timed region (§8d: dense)
    rag_h, _ = make_ragged(groups_per_rank, attempts=attempts, seq_length=seq_length, vocab=vocab,
                           seed=1234 + 2 + 1000 * rank, dense=True, with_ref=kl_coef > 0)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter()
    rag = rag_h.to(dev)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t_h2d
    rag_pinned = rag_h.pin_memory()
    torch.cuda.synchronize()
    t_h2d_pinned = time.perf_counter()
    rag = rag_pinned.to(dev)
    torch.cuda.synchronize()
    t_h2d_pinned = time.perf_counter() - t_h2d_pinned
    del rag_pinned
    h2d_bytes = sum(t.numel() * t.element_size() for t in (rag.tokens, rag.labels, rag.logprobs, rag.seq_off, rag.lp_off, rag.reward)
                    + ((rag.ref_logprobs,) if rag.ref_logprobs is not None else ()))
    n_seq = rag.n_seqs
    micro_batches = [[i] for i in range(n_seq)]  # dense: every sequence fills one seq_length budget
    tokens_per_rank = int(rag_h.host_seq_off[-1])

    # fp32 logits of one micro-batch (what the fp32 lm_head hands to the loss), generated once
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    logits = torch.empty((1, seq_length, vocab), dtype=torch.float32, device=dev)
    logits.normal_(0.0, 2.0, generator=gen)
    grad_logits = torch.empty_like(logits)

    # Make the rollouts on-policy w.r.t. these logits (untimed setup): old_logprobs = new_logprobs +
    # N(0, sigma) on completion tokens, so the importance ratio sits inside the PPO clip range and
    # EVERY completion row needs its full d-logits pass (rows whose gradient is exactly zero are
    # skipped by the kernel; sigma = 0.005 << epsilon = 0.02 keeps that shortcut out of the timing).
    from pipelinerl_amd.finetune.rl import logprob_entropy

    sigma = float(os.environ.get("PRL_BENCH_OLD_SIGMA", 0.005))
    setup = HotPathStep(cfg, eos_token_id=2)
    setup_batches = setup.preprocess(rag, micro_batches)
    lo = rag_h.host_lp_off
    for j in range(n_seq):
        b = setup_batches[j]
        nlp, _, _, _ = logprob_entropy(logits, b.input_ids, cfg.temperature)
        c = int(lo[j + 1] - lo[j])
        noise = torch.empty(c, device=dev).normal_(0.0, sigma, generator=gen)
        rag.logprobs[int(lo[j]) : int(lo[j + 1])] = nlp[0, seq_length - c :] + noise
        if rag.ref_logprobs is not None:  # KL-to-reference on: ref = old + N(0, 0.05) per SURVEY §8(d), a DIFFERENT column
            rag.ref_logprobs[int(lo[j]) : int(lo[j + 1])] = rag.logprobs[int(lo[j]) : int(lo[j + 1])] + torch.empty(c, device=dev).normal_(0.0, 0.05, generator=gen)
    del setup, setup_batches
    torch.cuda.synchronize()

    main_timer = EventTimer()
    grad_buckets = []
    if world > 1 and not args.no_grad_allreduce and args.backend == "nccl":
        grad_buckets = [torch.zeros(n // 2, dtype=torch.bfloat16, device=dev) for n in grad_bucket_sizes(args.grad_bytes)]

    def one_step(timed: bool, skip_unlabelled: bool = False, timer=None):
        # `value` is timed on the reference's behaviour: EVERY row of the logits is read, so that `isfinite(new_logprobs)` holds
        # over every position (rl/__init__.py:213); the opt-out that leaves unlabelled rows unread is timed afterwards, separately
        step = HotPathStep(cfg, eos_token_id=2, current_step=0, max_step=10, skip_unlabelled=skip_unlabelled)
        if timed:
            with timer.time("preprocess_K5_K6"):  # incl. the host planning (numpy plan + three small uploads)
                step.preprocess(rag, micro_batches, timer=timer)  # + event pairs around K5 and the K6 kernel alone
        else:
            step.preprocess(rag, micro_batches)
        for j in range(n_seq):
            if args.logits_mode == "fused":
                if timed:
                    with timer.time("fused_logits_loss"):
                        step.logits_backward(j, logits, grad_logits)
                else:
                    step.logits_backward(j, logits, grad_logits)
            else:
                step.logits_two_pass(j, logits, grad_logits)
        loss, stats = step.finish(timer=timer if timed else None)  # event pairs around the K2+K3 launch and around the cross-rank reduction
        if grad_buckets:  # the DP learner's exchange step: all-reduce of the gradients, bucket by bucket
            ctx = timer.time("grad_allreduce") if timed else None
            if ctx:
                ctx.__enter__()
            for gb in grad_buckets:
                dist.all_reduce(gb)
            if ctx:
                ctx.__exit__(None, None, None)
        return loss, stats

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, stats = one_step(True, timer=main_timer)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = 1e3 * elapsed / args.steps
    stats_host = stats.cpu().tolist()
    assert np.isfinite(stats_host[0]), "non-finite loss in the benchmark step"
    assert int(stats_host[2]) == bs, f"step covered {int(stats_host[2])} sequences, expected {bs}"

    kernels = main_timer.summary()
    V4 = vocab * 4
    labelled = int((rag.labels != -100).sum().item())
    live_frac = labelled / float(tokens_per_rank)

    def algorithmic_bytes(skip_unlabelled: bool) -> dict:
        """Algorithmic bytes per launch.  With every row read (the reference's behaviour) a row costs V*4 read + V*4
        written; with the opt-out the rows whose next token is unlabelled are not read, only their gradient row is zeroed."""
        read_frac = live_frac if skip_unlabelled else 1.0
        return {
            "fused_logits_loss": seq_length * ((1.0 + read_frac) * V4 + 56),  # logits read once + d logits written once
            "grpo_loss_step": tokens_per_rank * 52,                   # 56 B/token minus the unwritten 4 B gradient
            "preprocess_K5_K6": tokens_per_rank * (84 + 8),           # K6 16 B read + 68 B written; K5 scan 8 B read
            "pack_collate_kernel": tokens_per_rank * 84,              # the K6 kernel alone
            "group_advantages_K5": tokens_per_rank * 8,               # the K5 scan (+ O(S) group arithmetic), host planning + upload included
            "group_advantages_K5_kernels": tokens_per_rank * 8,       # the K5 scan launch alone (the O(S) group launch is timed next to it)
        }

    def price(table: dict, algo: dict) -> None:
        for name, k in table.items():
            if name in algo:
                k["algorithmic_bytes"] = algo[name]
                k["GBps"] = algo[name] / (k["avg_us"] * 1e-6) / 1e9
                k["hbm_frac"] = k["GBps"] / HBM_PEAK_GBS

    price(kernels, algorithmic_bytes(False))
    dom = "fused_logits_loss" if "fused_logits_loss" in kernels else max(kernels, key=lambda n: kernels[n]["avg_us"] * kernels[n]["launches"])

    # ---- the opt-out (HotPathStep's default, skip_unlabelled=True), timed separately on a few steps: never `value` ----
    optout = None
    if args.skip_unlabelled_steps > 0:
        t_opt = EventTimer()  # (no warm-up of its own: same kernels, same buffers, the flag is a launch argument)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.skip_unlabelled_steps):
            one_step(True, skip_unlabelled=True, timer=t_opt)
        barrier()
        dt_opt = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt_opt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_opt = t.item()
        k_opt = t_opt.summary()
        price(k_opt, algorithmic_bytes(True))
        optout = {"value": bs / (dt_opt / args.skip_unlabelled_steps), "steps": args.skip_unlabelled_steps, "ms_per_step": 1e3 * dt_opt / args.skip_unlabelled_steps,
                  "what": "HotPathStep(skip_unlabelled=True): rows whose next token carries no label are not read (no isfinite check there)",
                  "kernel": {k: k_opt[dom].get(k) for k in ("avg_us", "launches", "algorithmic_bytes", "GBps", "hbm_frac")} if dom in k_opt else None}

    traffic = None
    traffic_source = None
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    if pmc.exists() and (seq_length, vocab) == (8192, 152064):  # the PMC passes were taken at this shape
        try:
            table = json.loads(pmc.read_text())
            traffic = table.get(dom, {}).get("hbm_bytes_per_launch")
            traffic_source = "committed: profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; every row read)"
            # per-token PMC figures of the two step-scale kernels, scaled to this launch
            if "grpo_loss_step" in kernels and "hbm_bytes_per_token" in table.get("grpo_loss_step", {}):
                kernels["grpo_loss_step"]["traffic"] = table["grpo_loss_step"]["hbm_bytes_per_token"] * tokens_per_rank
            if "preprocess_K5_K6" in kernels and "write_kb" in table.get("pack_collate", {}):
                kernels["preprocess_K5_K6"]["pack_write_traffic"] = table["pack_collate"]["write_kb"] * 1024 / table["pack_collate"]["tokens"] * tokens_per_rank
        except Exception:
            traffic = None
    skipped: dict[str, str] = {}

    def fits(name: str, estimate_s: float) -> bool:
        """Optional legs of a DEFAULT run stay inside --budget-s (the driver's lease is shared with the tests and the smoke)."""
        if args.detail or time.perf_counter() - T_START + estimate_s <= args.budget_s:
            return True
        skipped[name] = f"not run: {time.perf_counter() - T_START:.0f} s spent, ~{estimate_s:.0f} s needed, budget {args.budget_s:.0f} s (--detail or --budget-s lifts it)"
        return False

    cpu_base = None if (args.no_cpu_baseline or world != 1) else cpu_baseline(seq_length, vocab)

    # The weight-sync probe creates its own RCCL communicator; a hang there must not cost the
    # benchmark line, so a watchdog prints the line without it and ends the process.
    full: dict = {}

    def release_step_buffers():
        nonlocal logits, grad_logits
        logits = grad_logits = None
        torch.cuda.empty_cache()

    wsync = None
    if world == 1 and not args.no_weight_sync and os.environ.get("PRL_BENCH_WSYNC", "1") != "0":
        # (never skipped for time: "trainer->actor weight-sync ms" is half of BASELINE.json's metric)  one GPU: the only trainer -> actor layout is colocated; hand the 7B / 0.5B parameter set to a
        # second process on this GPU over HIP IPC (request-to-ack of send_weight_update, median of 5)
        release_step_buffers()
        try:
            from pipelinerl_amd.weight_sync_probe import colocated_probe

            wsync = colocated_probe(param_set, iters=5, rehome=True, ready_timeout=240.0 if param_set != "32b" else 600.0)
            wsync = {"transport": "hip_ipc_colocated", **wsync,
                     "transport_note": "ONE GPU: trainer and inference worker share it, the bytes never touch a link - this figure says nothing about xGMI; the RCCL "
                                       "broadcast BASELINE's metric names needs >= 2 GPUs (`bench.py --gpus N` reports it under the same keys, transport rccl_xgmi)"}
        except Exception as e:  # noqa: BLE001 - the probe must never take the benchmark line down
            wsync = {"error": f"{type(e).__name__}: {e}"}

    if (rank == 0 and world == 1 and dom == "fused_logits_loss" and (seq_length, vocab) == (8192, 152064) and not args.no_live_pmc
            and os.environ.get("PRL_BENCH_LIVE_PMC", "1") != "0" and fits("live_pmc", 45)):
        release_step_buffers()
        try:
            live = live_pmc_traffic()
        except Exception:  # noqa: BLE001 - must never take the benchmark line down
            live = None
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            traffic_source = (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) over scripts/kernel_sweep.py --quick, "
                              f"{live['launches']} launches, every row read; FETCH_SIZE {live['fetch_kb']:.0f} KB x2 (gfx950) + WRITE_SIZE {live['write_kb']:.0f} KB")
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": kernels[dom]["hbm_frac"], "traffic": traffic, "traffic_source": traffic_source,
        # what `achieved` is made of, so that the line is self-sufficient: algorithmic bytes per launch / average launch duration
        "avg_us": kernels[dom]["avg_us"], "min_us": kernels[dom]["min_us"], "launches": kernels[dom]["launches"],
        "algorithmic_bytes_per_launch": kernels[dom].get("algorithmic_bytes"),
        "timing": "HIP events on torch's current stream (where the kernel is launched), every launch of the timed region",
    }
    if grad_buckets and "grad_allreduce" in kernels:
        k = kernels["grad_allreduce"]
        k["bytes"] = sum(b.numel() * 2 for b in grad_buckets)
        k["algbw_GBps"] = k["bytes"] / (k["avg_us"] * 1e-6) / 1e9
        k["busbw_GBps"] = 2 * (world - 1) / world * k["algbw_GBps"]  # ring all-reduce: every byte crosses 2 (N - 1) / N links

    # ---- side measurements (--detail only; they go to the detail file, never to the printed line) ----
    side: dict = {}
    if args.detail:
        release_step_buffers()
        if world == 1 and not args.no_fused_head:
            try:
                side["roofline_mfma"] = fused_head_probe(dev, seq_length, vocab, hidden)
            except Exception as e:  # noqa: BLE001 - must never take the benchmark line down
                side["roofline_mfma"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and param_set == "7b" and vocab == 152064:
            side["e2e"] = e2e_probe(live=not args.no_e2e and os.environ.get("PRL_BENCH_E2E", "1") != "0")
        if not args.no_ref_logprob and not args.no_fused_head:  # every rank runs it (no collective inside), rank 0 reports
            try:
                side["ref_logprob"] = ref_logprob_probe(dev, seq_length, vocab, hidden)
            except Exception as e:  # noqa: BLE001
                side["ref_logprob"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_preprocess_loop:
            try:
                side["preprocess_loop"] = preprocess_loop_probe(dev, seq_length, vocab, attempts)
            except Exception as e:  # noqa: BLE001
                side["preprocess_loop"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_pipeline and os.environ.get("PRL_BENCH_PIPELINE", "1") != "0":
            try:
                torch.cuda.empty_cache()
                side["pipeline"] = pipeline_probe(args.pipeline_steps, tiny=args.workload == "tiny")
            except Exception as e:  # noqa: BLE001 - must never take the benchmark line down
                side["pipeline"] = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0 and not args.no_transport:
            try:
                side["transport"] = transport_probe(seq_length, vocab)
            except Exception as e:  # noqa: BLE001
                side["transport"] = {"error": f"{type(e).__name__}: {e}"}

    label = {"7b_grpo_bs4096_seq8192": "7B GRPO bs=4096", "0p5b_grpo_bs512_seq2048": "0.5B GRPO bs=512 seq=2048",
             "32b_grpo_kl_bs4096_seq8192": "32B GRPO bs=4096, KL-to-ref on"}.get(args.workload, args.workload)

    def emit(wsync):
        if rank != 0:
            return
        full.update({
            "metric": f"learner samples/sec, {label} (post-model hot path on resident fp32 logits: K5+K6 preprocess, fused logits->GRPO loss->dlogits, "
                      "step stats; excludes the transformer forward/backward; trainer->actor weight-sync ms in weight_sync)",
            "value": bs / (elapsed / args.steps),
            "unit": "samples/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "global_batch": bs, "seq_len": seq_length, "vocab": vocab,
                       "tokens_per_step": bs * seq_length, "parallelism": f"dp{world}", "logits_mode": args.logits_mode,
                       "policy_loss": "ppo", "kl_coef": kl_coef, "skip_unlabelled": False, "labelled_token_fraction": live_frac,
                       "grad_allreduce_bytes_per_step": sum(b.numel() * 2 for b in grad_buckets) if grad_buckets else 0,
                       "backend": dist.get_backend() if world > 1 else None, "distinct_devices": distinct_devices},
            "config_detail": {"ref_logprobs": ("old + N(0, 0.05)" if kl_coef > 0 else "== old (KL off)"), "param_set": param_set, "head_hidden": hidden,
                              "old_logprob_sigma": sigma,
                              "torch_distributed": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices_visible": torch.cuda.device_count(),
                                                     "distinct_devices": distinct_devices, "rccl_comm_size": (wsync or {}).get("rccl_comm_size"),
                                                     "share_device_dry_run": share_device} if world > 1 else None),
                              "h2d_ragged_input": {"bytes": h2d_bytes, "ms": 1e3 * t_h2d, "ms_pinned": 1e3 * t_h2d_pinned,
                                                   "note": "one step's ragged rollouts, pageable vs page-locked host memory; not part of value"}},
            "roofline": roofline,
            "value_skip_unlabelled": optout,
            "kernels": kernels,
            "cpu_baseline": cpu_base,
            "weight_sync": wsync,
            "skipped": skipped,
            "loss": stats_host[0],
            "wall_s": time.perf_counter() - T_START,
            **side,
        })
        line, detail = split_line(full, args.detail_out)
        out = ROOT / args.detail_out
        try:
            out.parent.mkdir(parents=True, exist_ok=True)
            out.write_text(json.dumps(detail, indent=1) + "\n")
        except OSError as e:  # a read-only tree must not cost the line
            line["detail"] = f"not written: {e}"
        print(json.dumps(line), flush=True)

    force = os.environ.get("PRL_BENCH_FORCE_WSYNC") == "1"  # dry runs: exercise the probe's error handling under gloo
    if world > 1 and (args.backend == "nccl" or force) and not args.no_weight_sync and os.environ.get("PRL_BENCH_WSYNC", "1") != "0":
        import threading

        done = threading.Event()
        wsync = {"params": param_set, "param_bytes": grad_bytes_default, "transport": "rccl_xgmi", "metric": "trainer_to_actor_weight_sync_ms"}

        def watchdog():
            if not done.wait(float(os.environ.get("PRL_BENCH_WSYNC_TIMEOUT", 90 if param_set != "32b" else 300))):
                emit({**wsync, "error": f"weight-sync probe timed out in stage {wsync.get('stage')}"})
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        release_step_buffers()
        weight_sync_probe(rank, world, dev, wsync)
        done.set()
    emit(wsync)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
