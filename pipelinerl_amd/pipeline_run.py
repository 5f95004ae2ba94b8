"""The hot path AS a pipeline: the four stages of BASELINE `configs[1]` ("Qwen2.5-0.5B GRPO, 1 x MI355X actor + learner
colocated, synthetic rollouts bs=512 seq=2048") as four OS processes on one GPU, overlapping like the reference's
(`launch.py` fans the same stages out as processes; the stage bodies are finetune_loop.py:647-957, preprocess.py:370-704,
actor.py:510-557 / 648-652, vllm1.py:137-186):

    actor         ActorHarness over the rollout / dataset PLUGINS (synthetic_plugin) and a scripted llm; groups of `attempts`
                  rollouts as PRLROL01 records on the `actor` topic; paced by `max_lag` against the trainer's PROPAGATED
                  weight version (actor.py:510-557), stamped with that version (actor.py:210-219)
    preprocessor  PreprocessorLoop (shm streams, chunk_n_groups = 2): K5 on the device, the reference's scheduler, K6 per
                  drain, `training_data/0/0`; back-pressure from the trainer's `SamplesProcessed` (preprocess.py:587-592)
    learner       a random-init policy of the configured shape (Hugging Face Qwen2 layout) with `install_fused_head` (no
                  [T, V] logits), AdamW, `StreamedLearnerStep` (no host sync inside a step), loader thread; after every
                  optimizer step the weight_update_interval rule -> `WeightUpdateManager(transport="ipc")`: POST to the
                  engine, `WeightUpdateSuccess` to the trainer topic (finetune_loop.py:205-292, 936-949)
    engine        the inference-worker side: `StandaloneWeightReceiver` holding its own copy of the weights behind the
                  reference's update manager (pause(keep) -> collective_rpc -> resume, vllm1.py:137-186) and its one HTTP
                  route; optionally a scripted "generation" load on the GPU that the pause really stops

Every stage writes a report (`<exp_path>/reports/<stage>.json`: wall / busy seconds, queue gauges, per-update timings);
`run_pipeline` merges them into one object (bench.py `pipeline`, scripts/pipeline_cfg1.py).  Nothing here is measured
against the oracle or uses it: the parity of a pipelined step is tests/test_gpu_pipeline_procs.py.
"""

from __future__ import annotations

import json
import logging
import os
import time
import traceback
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any

logger = logging.getLogger(__name__)

MODEL_SHAPES = {
    # name: vocab, hidden, intermediate, layers, heads, kv heads, tied embeddings   (Qwen2.5 model cards; weight_sync_probe.qwen25_shapes)
    "0p5b": dict(vocab=151936, hidden=896, inter=4864, layers=24, heads=14, kv=2, tied=True),
    "7b": dict(vocab=152064, hidden=3584, inter=18944, layers=28, heads=28, kv=4, tied=False),
    "tiny": dict(vocab=512, hidden=64, inter=128, layers=2, heads=4, kv=2, tied=True),
}


@dataclass
class PipelineSpec:
    exp_path: str
    model: str = "0p5b"
    global_batch: int = 512            # samples per optimizer step (train_batch_size 1 x gradient_accumulation_passes, App. E)
    seq_length: int = 2048             # longest rollout (prompt + completion)
    pack_budget: int | None = None     # tokens per packed micro-batch (`finetune.seq_length`); default: seq_length, the reference's coupling
    attempts: int = 8
    steps: int = 5
    vocab: int | None = None           # default: the model's
    max_lag: int | None = None         # samples; None = one optimizer step's worth (the actor may run one step ahead)
    weight_update_interval: int = 1
    chunk_n_groups: int = 2
    dense: bool = False                # every rollout exactly seq_length tokens (SURVEY §8d worst case) instead of ragged
    seed: int = 1235                   # 1234 + config index 1
    lr: float = 1e-6
    n_problems: int = 64
    concurrent_groups: int = 4
    engine_load: bool = False          # the engine runs forward passes between updates (a colocated actor's GPU share)
    gradient_checkpointing: bool = False
    learner: str = "streamed"          # "streamed" (StreamedLearnerStep) or "dropin" (LearnerStep + rl_step_fused_head)
    optimizer: str = "adamw"           # "adamw" | "sgd" (parity tests)
    param_dtype: str = "bf16"          # "bf16" (the reference's training dtype) | "fp32" (parity tests: tight parameter deltas)
    wire: str = "full"                 # `training_data` records: "full" (the reference's expanded batch) | "compact" (ragged columns; K6 on the learner's GPU)
    mirror_jsonl: bool = False         # JSONL mirrors of `actor` and `training_data` (replay / parity tests; compact wire: `actor` only)
    retain_streams: bool = False       # keep consumed segments of the bulk topics (isolated-stage reruns read them again)
    capture_step0: str | None = None   # directory: the learner saves step 0's micro-batches and the parameters around it
    segment_mb: int = 64
    stage_timeout_s: float = 900.0
    stacks_after_s: float = 0.0        # diagnosis: every stage still alive after this many seconds dumps its threads' Python stacks
    device: int = 0
    extra: dict = field(default_factory=dict)

    @property
    def shape(self) -> dict:
        s = dict(MODEL_SHAPES[self.model])
        if self.vocab:
            s["vocab"] = int(self.vocab)
        return s

    @property
    def budget(self) -> int:
        return int(self.pack_budget or self.seq_length)

    @property
    def lag(self) -> int:
        return self.global_batch if self.max_lag is None else int(self.max_lag)


def _set_backend(spec: PipelineSpec, owner: bool = False) -> None:
    from . import streams

    streams.reset_streams_backend()
    opts: dict[str, Any] = {"segment_bytes": spec.segment_mb << 20, "owner": owner}
    if spec.retain_streams:
        opts["trim_topics"] = ()
    if spec.mirror_jsonl:
        opts["mirror_jsonl"] = ["actor"] if spec.wire == "compact" else ["actor", "training_data"]
    streams.set_streams_backend("shm", **opts)


def _trace(spec: PipelineSpec, stage: str, what: str) -> None:
    """Start-up milestones of a stage, appended to `<exp>/reports/<stage>.trace` (a run that times out before any stage has
    reported says where each stage was)."""
    d = Path(spec.exp_path) / "reports"
    d.mkdir(parents=True, exist_ok=True)
    with open(d / f"{stage}.trace", "a") as f:
        f.write(f"{time.time():.3f} {what}\n")


def _report(spec: PipelineSpec, stage: str, data: dict) -> None:
    d = Path(spec.exp_path) / "reports"
    d.mkdir(parents=True, exist_ok=True)
    tmp = d / f".{stage}.json.tmp"
    tmp.write_text(json.dumps(data))
    tmp.rename(d / f"{stage}.json")


def _stage(fn):
    """Stage entry point: spec dict in, a report out - an error report (with the traceback) when the stage dies."""

    def main(spec_dict: dict) -> None:
        spec = PipelineSpec(**spec_dict)
        logging.basicConfig(level=os.environ.get("PRL_PIPELINE_LOG", "WARNING"), format=f"%(asctime)s {fn.__name__} %(levelname)s %(message)s")
        name = fn.__name__.replace("_stage", "")
        if spec.stacks_after_s:  # a stage that is still running then leaves the Python stacks of all its threads behind
            import faulthandler

            d = Path(spec.exp_path) / "reports"
            d.mkdir(parents=True, exist_ok=True)
            faulthandler.dump_traceback_later(spec.stacks_after_s, repeat=False, file=open(d / f"{name}.stacks", "w"), exit=False)
        try:
            fn(spec)
        except BaseException:  # noqa: BLE001 - the orchestrator reads it
            _report(spec, fn.__name__.replace("_stage", ""), {"error": traceback.format_exc()})
            raise

    main.__name__ = fn.__name__
    main.__qualname__ = fn.__qualname__
    return main


def rl_config_of(spec: PipelineSpec):
    """conf/finetune/grpo.yaml over base.yaml:100-114."""
    from .finetune.rl import RLConfig

    return RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, clamp_log_ratio_ref_new_value=5,
                    temperature=1.0, divide_advantage_by_std=False, group_normalization=False, batch_size=spec.global_batch)


def build_policy(spec: PipelineSpec, device, seed: int):
    """A random-init causal LM of the configured Qwen2.5 shape, bf16, Hugging Face layout (`.model` + `.lm_head`), built
    directly in HBM.  A tied head keeps its bf16 storage like the reference's `apply_fp32_lm_head` leaves it
    (finetune/checkpoints.py:74-83); the fused head reads it as one exact plane."""
    import torch
    import transformers

    s = spec.shape
    cfg = transformers.Qwen2Config(vocab_size=s["vocab"], hidden_size=s["hidden"], intermediate_size=s["inter"], num_hidden_layers=s["layers"],
                                   num_attention_heads=s["heads"], num_key_value_heads=s["kv"], max_position_embeddings=max(32768, spec.budget),
                                   tie_word_embeddings=s["tied"], attn_implementation="sdpa")
    torch.manual_seed(seed)
    with torch.device(device):
        torch.set_default_dtype(torch.bfloat16)
        try:
            model = transformers.Qwen2ForCausalLM(cfg)
        finally:
            torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16 if spec.param_dtype == "bf16" else torch.float32)
    if not s["tied"]:
        model.lm_head = model.lm_head.float()  # fp32 output head (checkpoints.py:87-103)
    return model


def _param_probe(named_parameters) -> dict[str, float]:
    """A cheap fingerprint of a parameter set (three tensors, fp64 sums): engine and trainer must agree on it after an update."""
    import torch

    params = dict(named_parameters)
    names = [n for n in params if n.endswith("norm.weight")][-1:] + [n for n in params if "layers.0.self_attn.q_proj.weight" in n][:1] \
        + [n for n in params if "embed_tokens.weight" in n][:1]
    with torch.no_grad():
        return {n: float(params[n].detach().double().sum().item()) for n in names}


# ---------------------------------------------------------------------------------------------
# stages
# ---------------------------------------------------------------------------------------------


@_stage
def actor_stage(spec: PipelineSpec) -> None:
    from .actor_harness import ActorHarness
    from .state import TrainerState
    from .synthetic_plugin import SyntheticLLM

    _set_backend(spec)
    state = TrainerState(Path(spec.exp_path))
    state.start_listening()
    cfg = {
        "attempts": spec.attempts,
        "actor": {"rollout_policy": "pipelinerl_amd.synthetic_plugin.generate_rollout"},
        "dataset_loader": "pipelinerl_amd.synthetic_plugin.load_problems",
        "dataset_loader_params": {"n_problems": spec.n_problems, "seed": spec.seed},
        "train_dataset_names": ["synthetic"],
    }
    llm = SyntheticLLM(spec.shape["vocab"], spec.seq_length, dense=spec.dense, prompt_max=min(512, max(8, spec.seq_length // 4)),
                       prompt_min=min(64, max(2, spec.seq_length // 32)))
    harness = ActorHarness(cfg, [llm], spec.exp_path, trainer_state=state, scheduler_name="actor0", wire="ragged", shuffle_seed=spec.seed)
    versions: list[int] = []
    t0 = time.perf_counter()
    n = harness.run_paced(samples_target=spec.steps * spec.global_batch, train_batch_size=1, gradient_accumulation_passes=spec.global_batch,
                          weight_update_interval=spec.weight_update_interval, max_lag=spec.lag, concurrent_groups=spec.concurrent_groups,
                          on_group=lambda g: versions.append(int(g[0].model_version)))
    t = harness.timing
    hist: dict[int, int] = {}
    for v in versions:
        hist[v] = hist.get(v, 0) + 1
    _report(spec, "actor", {"published_samples": n, "published_groups": harness.published_groups, "wall_s": time.perf_counter() - t0,
                            "busy_s": t["busy_s"], "blocked_by_lag_s": t["blocked_by_lag_s"], "busy_frac": t["busy_s"] / max(t["wall_s"], 1e-9),
                            "groups_per_model_version": {str(k): v for k, v in sorted(hist.items())}, "llm_calls": llm.calls,
                            "pacing": {"max_lag_samples": spec.lag, "weight_update_interval": spec.weight_update_interval,
                                       "budget": ActorHarness.submission_budget(spec.attempts, 1, spec.global_batch, spec.weight_update_interval, spec.lag)}})


@_stage
def preprocessor_stage(spec: PipelineSpec) -> None:
    import torch

    from .preprocess import PreprocessorConfig, PreprocessorLoop
    from .state import TrainerState

    _set_backend(spec)
    dev = torch.device("cuda", spec.device)
    torch.cuda.set_device(dev)
    state = TrainerState(Path(spec.exp_path))
    state.start_listening()
    state.wait_for_processed_samples()  # the trainer's first message (finetune_loop.py:462-465)
    cfg = PreprocessorConfig(exp_path=Path(spec.exp_path), num_trainers=1, train_batch_size=1, gradient_accumulation_passes=spec.global_batch,
                             seq_length=spec.budget, attempts=spec.attempts, rl=rl_config_of(spec), eos_token_id=2, chunk_n_groups=spec.chunk_n_groups,
                             max_lag=spec.lag, samples_target=spec.steps * spec.global_batch,
                             ring_buffer_size=max(128, 2 * spec.global_batch), max_ready_samples_per_lead=max(64, spec.global_batch))
    loop = PreprocessorLoop(cfg, dev, trainer_state=state, profile=True, wire=spec.wire)
    t0 = time.perf_counter()
    n = loop.run(idle_timeout=spec.stage_timeout_s)
    wall = time.perf_counter() - t0
    prof = dict(loop.prof or {})
    kern = loop.kernel_seconds()
    busy = sum(v for k, v in prof.items() if k != "input_wait")
    g = loop.gauges

    def pct(col: int, q: float) -> float:
        xs = sorted(r[col] for r in g)
        return float(xs[min(len(xs) - 1, int(q * len(xs)))]) if xs else 0.0

    _report(spec, "preprocessor", {"published_samples": n, "wall_s": wall, "busy_s": busy, "busy_frac": busy / max(wall, 1e-9),
                                   "host_phase_s": prof, "kernel_s": kern, "backpressure_waits": loop.backpressure_waits,
                                   "chunks": loop._next_chunk,
                                   "queue_depth": {"raw_chunks": {"median": pct(1, 0.5), "p90": pct(1, 0.9), "max": pct(1, 1.0)},
                                                   "ring_plus_buffer_samples": {"median": pct(2, 0.5), "p90": pct(2, 0.9), "max": pct(2, 1.0)},
                                                   "published_not_yet_trained_samples": {"median": pct(3, 0.5), "p90": pct(3, 0.9), "max": pct(3, 1.0)},
                                                   "gauge_samples": len(g)}})


@_stage
def engine_stage(spec: PipelineSpec) -> None:
    import torch

    from .engine_update import InflightUpdateManager, ScriptedEngine, UpdateServer
    from .state import TrainerState
    from .vllm_worker import StandaloneWeightReceiver

    _set_backend(spec)
    dev = torch.device("cuda", spec.device)
    torch.cuda.set_device(dev)
    _trace(spec, "engine", "building the policy")
    model = build_policy(spec, dev, seed=spec.seed + 999)  # different values than the trainer's: an update must really land
    _trace(spec, "engine", "policy built")
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    worker = StandaloneWeightReceiver(model, dev)
    gen_tokens = min(256, spec.seq_length)
    ids = torch.randint(3, spec.shape["vocab"], (4, gen_tokens), device=dev)

    def generate_step():
        with torch.no_grad():
            model.model(input_ids=ids)
        torch.cuda.synchronize(dev)

    engine = ScriptedEngine([worker], generate_step if spec.engine_load else None)
    manager = InflightUpdateManager(engine)
    server = UpdateServer(manager)
    (Path(spec.exp_path) / "reports").mkdir(parents=True, exist_ok=True)
    (Path(spec.exp_path) / "reports" / "engine_url.txt").write_text(server.url)
    _trace(spec, "engine", "serving " + server.url)
    state = TrainerState(Path(spec.exp_path))
    state.start_listening()
    t0 = time.perf_counter()
    state.wait_for_training_done(timeout=spec.stage_timeout_s)
    wall = time.perf_counter() - t0
    time.sleep(0.2)  # a POST that raced the TrainingDone message finishes
    probe = _param_probe(model.named_parameters())
    tm = manager.timings
    med = lambda k: sorted(t[k] for t in tm)[len(tm) // 2] if tm else None  # noqa: E731
    _report(spec, "engine", {"updates": len(tm), "wall_s": wall, "last_version": tm[-1]["version"] if tm else None, "param_probe": probe,
                             "pause_ms_median": 1e3 * med("pause_s") if tm else None, "update_ms_median": 1e3 * med("update_s") if tm else None,
                             "resume_ms_median": 1e3 * med("resume_s") if tm else None, "per_update": tm,
                             "busy_s": sum(t["total_s"] for t in tm), "busy_frac": sum(t["total_s"] for t in tm) / max(wall, 1e-9),
                             "generation_quanta": engine.quanta, "generation_quanta_by_version": {str(k): v for k, v in engine.quanta_by_version.items()},
                             "engine_load": spec.engine_load})
    engine.shutdown()
    worker.close_communicator()
    server.close()


@_stage
def learner_stage(spec: PipelineSpec) -> None:
    import queue
    import threading

    import torch

    from . import streams
    from .finetune_loop import (TRAINER_TOPIC, LearnerStep, SamplesProcessed, StreamedLearnerStep, WeightUpdateManager, run_data_loader)
    from .fused_head import install_fused_head, rl_step_fused_head
    from .weight_sync import ColocatedSender

    _set_backend(spec)
    dev = torch.device("cuda", spec.device)
    torch.cuda.set_device(dev)
    t_init = time.perf_counter()
    _trace(spec, "learner", "building the policy")
    model = build_policy(spec, dev, seed=spec.seed)
    _trace(spec, "learner", "policy built")
    install_fused_head(model)
    if spec.gradient_checkpointing:
        model.gradient_checkpointing_enable()
    model.train()
    n_params = sum(p.numel() for p in model.parameters())
    param_bytes = sum(p.numel() * p.element_size() for p in model.parameters())
    # wait for the inference server like the reference does (finetune_loop.py:470)
    url_file = Path(spec.exp_path) / "reports" / "engine_url.txt"
    deadline = time.time() + spec.stage_timeout_s
    while not url_file.exists():
        if time.time() > deadline:
            raise TimeoutError("the engine never announced its url")
        time.sleep(0.05)
    url = url_file.read_text().strip()
    _trace(spec, "learner", "engine found at " + url)
    topic = streams.SingleStreamSpec(exp_path=Path(spec.exp_path), topic=TRAINER_TOPIC)
    mgr = WeightUpdateManager(llm_urls=[url], accelerated_model=model, update_stream=topic, actor_update_group=None, transport="ipc")
    mgr._sender = ColocatedSender(dev, mgr.bucket_bytes)
    mgr._sender.rehome(model.named_parameters())  # the parameters LIVE in the exported buckets: publishing an update copies nothing
    _trace(spec, "learner", "parameters rehomed into the exported buckets")
    if spec.optimizer == "sgd":
        opt = torch.optim.SGD(model.parameters(), lr=spec.lr)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=spec.lr, fused=True)
    rl = rl_config_of(spec)
    common = dict(train_batch_size=1, gradient_accumulation_passes=spec.global_batch, max_train_steps=spec.steps, weight_update_manager=mgr,
                  weight_update_interval=spec.weight_update_interval, trainer_stream=topic, max_lag=spec.lag)
    if spec.learner == "streamed":
        step = StreamedLearnerStep(model, opt, rl, **common)
    else:
        step = LearnerStep(model, opt, rl, rl_step_fn=rl_step_fused_head, **common)
    init_s = time.perf_counter() - t_init

    capture = Path(spec.capture_step0) if spec.capture_step0 else None
    if capture is not None:
        capture.mkdir(parents=True, exist_ok=True)
        torch.save({n: p.detach().cpu().clone() for n, p in model.named_parameters()}, capture / "params_before.pt")
        first_step = opt.step

        def step_and_keep_gradients(*a_, **k_):  # step 0's accumulated gradients, as the optimizer sees them
            torch.save({n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}, capture / "grads_step0.pt")
            opt.step = first_step
            return first_step(*a_, **k_)

        opt.step = step_and_keep_gradients
    captured: list = []

    # the trainer's first two messages (finetune_loop.py:462-485): where it stands, and the first weight version
    step.publish(SamplesProcessed(samples_processed=step.metrics.samples))
    sync_ms: list[float] = []
    t0 = time.perf_counter()
    _trace(spec, "learner", "sending weight version 0")
    mgr.send_weight_update(step.metrics.samples)
    first_sync_ms = 1e3 * (time.perf_counter() - t0)
    _trace(spec, "learner", f"weight version 0 acknowledged after {first_sync_ms:.0f} ms")
    probes = {str(step.metrics.samples): _param_probe(model.named_parameters())}

    q: queue.Queue = queue.Queue(maxsize=8)
    stop = threading.Event()
    data_spec = streams.SingleStreamSpec(exp_path=Path(spec.exp_path), topic="training_data", partition=0)
    threading.Thread(target=run_data_loader, args=(data_spec, q, dev, stop), kwargs={"annotate": spec.learner == "streamed"},
                     name="learner-loader", daemon=True).start()

    wait_s = 0.0
    lags: list[int] = []
    depth: list[int] = []
    step_marks: list[dict] = []
    micro_batches = tokens = 0
    t_loop = time.perf_counter()
    t_step = t_loop
    wait_step = 0.0
    last_loss = None
    while step.metrics.completed_steps < spec.steps:
        t0 = time.perf_counter()
        depth.append(q.qsize())
        while True:
            try:
                batch = q.get(timeout=1.0)
                break
            except queue.Empty:
                if time.perf_counter() - t0 > spec.stage_timeout_s:
                    raise TimeoutError("no training data arrived") from None
        if isinstance(batch, Exception):
            raise batch
        w = time.perf_counter() - t0
        wait_s += w
        wait_step += w
        if capture is not None and step.metrics.completed_steps == 0:
            captured.append({k: v.detach().cpu().clone() for k, v in batch.tensors()} | {"model_version": batch.model_version, "sentinel": batch.sentinel,
                                                                                            "padding": batch.padding, "is_packed": batch.is_packed})
        if not batch.sentinel:
            lags.append(step.metrics.samples - int(batch.model_version))
        res = step.step(batch)
        micro_batches += 1
        tokens += int(batch.input_ids.numel())
        if res["did_optimizer_step"]:
            last_loss = res["metrics"].get("rl/loss")
            if capture is not None and step.metrics.completed_steps == 1:
                torch.cuda.synchronize(dev)
                torch.save(captured, capture / "step0_batches.pt")
                torch.save({n: p.detach().cpu().clone() for n, p in model.named_parameters()}, capture / "params_after.pt")
                (capture / "step0_metrics.json").write_text(json.dumps(res["metrics"]))
            torch.cuda.synchronize(dev)
            t_opt = time.perf_counter()
            t1 = time.perf_counter()
            sent = step.maybe_send_weights()
            if sent:
                sync_ms.append(1e3 * (time.perf_counter() - t1))
                probes[str(step.metrics.samples)] = _param_probe(model.named_parameters())
            now = time.perf_counter()
            step_marks.append({"step": step.metrics.completed_steps, "wall_s": now - t_step, "waiting_for_data_s": wait_step,
                               "weight_sync_ms": sync_ms[-1] if sent else None, "compute_s": t_opt - t_step - wait_step, "loss": last_loss})
            t_step, wait_step = now, 0.0
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t_loop
    stop.set()
    step.finish()
    mgr.shutdown()
    hist: dict[int, int] = {}
    for x in lags:
        k = int(x) // spec.global_batch  # in optimizer steps
        hist[k] = hist.get(k, 0) + 1
    steady = step_marks[1:] if len(step_marks) > 1 else step_marks
    sync_sorted = sorted(sync_ms)
    _report(spec, "learner", {
        "completed_steps": step.metrics.completed_steps, "samples": step.metrics.samples, "micro_batches": micro_batches, "tokens": tokens,
        "wall_s": wall, "waiting_for_data_s": wait_s, "busy_s": wall - wait_s, "busy_frac": (wall - wait_s) / max(wall, 1e-9),
        "init_s": init_s, "params": n_params, "param_bytes": param_bytes,
        "steady_state": {"steps": len(steady), "s_per_step": sum(m["wall_s"] for m in steady) / max(len(steady), 1),
                         "samples_per_s": spec.global_batch * len(steady) / max(sum(m["wall_s"] for m in steady), 1e-9)},
        "per_step": step_marks,
        "weight_sync": {"transport": "hip_ipc_colocated", "first_ms": first_sync_ms, "under_load_ms": sync_ms,
                        "median_ms": sync_sorted[len(sync_sorted) // 2] if sync_sorted else None, "max_ms": sync_sorted[-1] if sync_sorted else None,
                        "what": "send_weight_update request -> engine paused, weights copied, resumed -> HTTP ack -> WeightUpdateSuccess, while the "
                                "preprocessor's kernels and the loader's copies keep running on the same GPU"},
        "batch_queue_depth": {"median": sorted(depth)[len(depth) // 2] if depth else 0, "max": max(depth) if depth else 0, "maxsize": 8},
        "lag_optimizer_steps_histogram": {str(k): v for k, v in sorted(hist.items())},
        "lag_what": "per micro-batch: (samples trained when it is consumed - model_version stamped on its oldest rollout) // samples per step",
        "samples_too_old_to_train": step.metrics.samples_too_old_to_train, "param_probes": probes, "final_loss": last_loss, "learner": spec.learner,
        "peak_memory_GB": torch.cuda.max_memory_allocated(dev) / 1e9})
    # (the exported buckets are not freed here: the engine may still have them mapped - they go with the process)


# ---------------------------------------------------------------------------------------------
# orchestration
# ---------------------------------------------------------------------------------------------

STAGES = {"engine": engine_stage, "learner": learner_stage, "preprocessor": preprocessor_stage, "actor": actor_stage}


def run_pipeline(spec: PipelineSpec, timeout_s: float | None = None) -> dict:
    """Run the four stages as processes until the learner has done `spec.steps` optimizer steps; returns the merged report."""
    import multiprocessing as mp
    import shutil

    from . import streams

    exp = Path(spec.exp_path)
    if (exp / "reports").exists():
        shutil.rmtree(exp / "reports")
    exp.mkdir(parents=True, exist_ok=True)
    was = (streams._backend, dict(streams._backend_options))
    _set_backend(spec, owner=True)
    streams.begin_run(exp)
    ctx = mp.get_context("spawn")
    procs = {}
    t0 = time.perf_counter()
    timeout_s = timeout_s or spec.stage_timeout_s
    try:
        for name in ("engine", "learner", "preprocessor", "actor"):
            procs[name] = ctx.Process(target=STAGES[name], args=(asdict(spec),), name=f"prl-{name}", daemon=True)
            procs[name].start()
        failed = None
        while time.perf_counter() - t0 < timeout_s:
            alive = {n: p.is_alive() for n, p in procs.items()}
            dead_bad = [n for n, p in procs.items() if not p.is_alive() and p.exitcode not in (0, None)]
            if dead_bad:
                failed = dead_bad[0]
                break
            if not any(alive.values()):
                break
            time.sleep(0.05)
        wall = time.perf_counter() - t0
        timed_out = any(p.is_alive() for p in procs.values())
    finally:
        for p in procs.values():
            if p.is_alive():
                p.terminate()
        for p in procs.values():
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    reports = {}
    for name in STAGES:
        f = exp / "reports" / f"{name}.json"
        reports[name] = json.loads(f.read_text()) if f.exists() else {"error": "no report"}
    streams.clean_shm_streams(exp)
    streams.reset_streams_backend()
    if was[0] is not None:
        streams.set_streams_backend(was[0], **was[1])
    errors = {n: r["error"] for n, r in reports.items() if "error" in r}
    out: dict[str, Any] = {"spec": {k: v for k, v in asdict(spec).items() if k not in ("extra",)}, "wall_s_incl_start_up": wall, "stages": reports}
    if failed or timed_out or errors:
        traces = {}
        for name in STAGES:
            f = exp / "reports" / f"{name}.trace"
            if f.exists():
                lines = f.read_text().splitlines()
                t0 = float(lines[0].split()[0]) if lines else 0.0
                traces[name] = [f"+{float(x.split()[0]) - t0:.1f}s {' '.join(x.split()[1:])}" for x in lines]
        stacks = {n: (exp / "reports" / f"{n}.stacks").read_text()[-6000:] for n in STAGES if (exp / "reports" / f"{n}.stacks").exists()
                  and (exp / "reports" / f"{n}.stacks").stat().st_size}
        out["error"] = {"failed_stage": failed, "timed_out": timed_out, "stage_errors": {n: e[-1500:] for n, e in errors.items()}, "start_up_traces": traces,
                        "stacks": stacks}
        return out
    out["summary"] = summarize(spec, reports)
    return out


def summarize(spec: PipelineSpec, r: dict) -> dict:
    """The numbers the bench line quotes: steady-state samples/s, who was busy, what waited, weight sync under load, lag."""
    L, P, A, E = r["learner"], r["preprocessor"], r["actor"], r["engine"]
    steps = max(L["completed_steps"], 1)
    busy_per_step = {"actor_s": A["busy_s"] / steps, "preprocessor_s": P["busy_s"] / steps, "learner_s": L["busy_s"] / steps,
                     "engine_s": E["busy_s"] / max(E["updates"], 1)}
    probes_agree = None
    last = str(E.get("last_version"))
    if last in L.get("param_probes", {}):
        probes_agree = L["param_probes"][last] == E["param_probe"]
    return {
        "samples_per_s": L["steady_state"]["samples_per_s"], "s_per_step": L["steady_state"]["s_per_step"], "steady_state_steps": L["steady_state"]["steps"],
        "tokens_per_s": L["tokens"] / max(L["wall_s"], 1e-9), "optimizer_steps": L["completed_steps"],
        "busy_frac": {"actor": A["busy_frac"], "preprocessor": P["busy_frac"], "learner": L["busy_frac"], "engine": E["busy_frac"]},
        "stage_busy_s_per_step": busy_per_step,
        "sum_of_stage_busy_s_per_step": sum(busy_per_step.values()),
        "overlap": {"pipelined_s_per_step": L["wall_s"] / steps, "stages_back_to_back_s_per_step": sum(busy_per_step.values()),
                    "what": "busy seconds of every stage per optimizer step, measured inside the running pipeline, added up (= the step time if the stages "
                            "ran one after the other) next to the wall time of a step with the stages overlapping"},
        "queue_depth": {"preprocessor": P["queue_depth"], "learner_batch_queue": L["batch_queue_depth"]},
        "weight_sync_under_load_ms": {"median": L["weight_sync"]["median_ms"], "max": L["weight_sync"]["max_ms"], "first": L["weight_sync"]["first_ms"],
                                       "updates": len(L["weight_sync"]["under_load_ms"]), "transport": "hip_ipc_colocated",
                                       "engine_pause_update_resume_ms": [E["pause_ms_median"], E["update_ms_median"], E["resume_ms_median"]]},
        "lag_optimizer_steps_histogram": L["lag_optimizer_steps_histogram"],
        "actor_blocked_by_lag_s": A["blocked_by_lag_s"], "preprocessor_backpressure_waits": P["backpressure_waits"],
        "engine_weights_equal_trainer_at_last_version": probes_agree, "final_loss": L["final_loss"], "learner_peak_memory_GB": L["peak_memory_GB"],
    }
