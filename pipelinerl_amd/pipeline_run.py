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

The same stages run as **N learners x M engines** (BASELINE configs[2] / [3]: 2 + 2 and 4 + 4; `PipelineSpec(n_learners=, n_engines=,
weight_transport=)`): the preprocessor publishes to `training_data/0/{rank}` for N lead trainers with the per-step quota and the
sentinel rule (preprocess.py:462-481, 596-662), the N learner ranks form one torch.distributed group (DDP; RCCL when every rank has
its own GPU, gloo when they share one) and keep the reference's per-rank sample accounting (finetune_loop.py:627-646, 709), every rank
calls `send_weight_update` and rank 0 sends (finetune_loop.py:205-292) - to a weight-update group of M + 1 members over RCCL
(`weight_transport="rccl"`, one GPU per member), over gloo (`"gloo"`: the same group and protocol where RCCL cannot run) or through HIP
IPC handles every colocated engine maps (`"ipc"`) - and every engine acknowledges before `WeightUpdateSuccess` is published.  GPU
placement follows world.py:143-192 (inference GPUs first, learners after them) unless `share_device` puts every stage on one GPU.
`engine_tp > 1` (configs[4]: TP = 2 engines): an engine is `tp` inference workers holding vLLM-style stacked slices; the trainer forms
one weight-update group PER TP RANK (rank layout vllm1.py:71) and every worker receives only its slices (`transport: sharded`,
tp_shard.py).  `kl_coef > 0` puts the frozen reference policy into the preprocessor.  `baseline_spec(k)` builds configs[1..4].

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
    "32b": dict(vocab=152064, hidden=5120, inter=27648, layers=64, heads=40, kv=8, tied=False),
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
    kl_coef: float = 0.0               # > 0 (configs[4]: 0.001): KL-to-reference on - the preprocessor holds the frozen reference policy (the initial policy) and writes `ref_logprobs`
    ref_seed: int | None = None        # seed of the reference policy (default: `seed`, i.e. the policy the run starts from)
    wire: str = "full"                 # `training_data` records: "full" (the reference's expanded batch) | "compact" (ragged columns; K6 on the learner's GPU)
    mirror_jsonl: bool = False         # JSONL mirrors of `actor` and `training_data` (replay / parity tests; compact wire: `actor` only)
    retain_streams: bool = False       # keep consumed segments of the bulk topics (isolated-stage reruns read them again)
    capture_step0: str | None = None   # directory: the learner saves step 0's micro-batches and the parameters around it
    segment_mb: int = 64
    stage_timeout_s: float = 900.0
    stacks_after_s: float = 0.0        # diagnosis: every stage still alive after this many seconds dumps its threads' Python stacks
    device: int = 0
    n_learners: int = 1                # data-parallel learner ranks (lead trainers; seq_parallel = 1)
    n_engines: int = 1                 # inference engines, `engine_tp` GPUs (weight-group members) each
    engine_tp: int = 1                 # tensor-parallel degree of an engine: > 1 = every TP rank receives only ITS slices (`transport: sharded`, tp_shard.py)
    weight_transport: str = "ipc"      # "ipc" (HIP IPC handles, engines colocated with learner 0) | "rccl" | "gloo" (group of n_engines + 1)
    grad_backend: str | None = None    # learners' process group: default "nccl" with one GPU per learner, "gloo" when they share a device / on CPU
    share_device: bool = True          # every stage on `device`; False: engines on GPUs [device, device + M), learners on the N after them (world.py:143-192)
    platform: str = "cuda"             # "cpu": host tensors - topology tests; needs `hooks` (the loss and preprocessing kernels have no CPU path)
    hooks: str | None = None           # module with optional `build_policy(spec, device, seed)`, `rl_step_fn`, `preprocessor_stage(spec)` replacements
    sys_path: list = field(default_factory=list)  # prepended to sys.path in every stage process (where `hooks` lives)
    learner_port: int = 0              # rendezvous ports on 127.0.0.1, chosen by run_pipeline
    wsync_port: int = 0
    extra: dict = field(default_factory=dict)

    def __post_init__(self):
        if self.n_learners < 1 or self.n_engines < 1:
            raise ValueError("a pipeline has at least one learner and one engine")
        if self.global_batch % self.n_learners:
            raise ValueError(f"global_batch {self.global_batch} does not split into {self.n_learners} equal per-learner quotas "
                             "(the launcher rounds gradient_accumulation_passes up to a multiple of the learner count, launch.py:631-640)")
        if self.weight_transport not in ("ipc", "rccl", "gloo"):
            raise ValueError(f"weight_transport {self.weight_transport!r}: 'ipc', 'rccl' or 'gloo'")
        if self.weight_transport == "ipc" and not self.share_device:
            raise ValueError("weight_transport 'ipc' hands HIP IPC handles to engines on the learner's own GPU: it needs share_device=True")
        if self.weight_transport == "rccl" and self.share_device:
            raise ValueError("weight_transport 'rccl' needs one GPU per member of the weight-update group (RCCL refuses two ranks on one device): "
                             "share_device=False, or 'gloo' / 'ipc' on a shared device")
        if self.engine_tp < 1 or (self.engine_tp > 1 and self.weight_transport == "ipc"):
            raise ValueError("engine_tp > 1 is the sharded update over a per-TP-rank group: weight_transport 'rccl' or 'gloo'")
        if self.platform not in ("cuda", "cpu"):
            raise ValueError(f"platform {self.platform!r}")
        if self.platform == "cpu" and (self.weight_transport != "gloo" or not self.hooks):
            raise ValueError("platform 'cpu' runs the topology with host tensors: weight_transport='gloo' and a `hooks` module are required")

    @property
    def learner_backend(self) -> str:
        if self.grad_backend:
            return self.grad_backend
        return "nccl" if (self.platform == "cuda" and not self.share_device) else "gloo"

    def device_of(self, stage: str, index: int = 0, tp_rank: int = 0):
        """The torch device of a stage process (of TP rank `tp_rank` of engine `index`).  Not shared: inference GPUs first (engine e, TP
        rank t on GPU e * engine_tp + t), learner GPUs after them - the order in which the reference's WorldMap hands out a node's GPUs
        (world.py:143-192); the preprocessor computes on the first learner's GPU."""
        import torch

        if self.platform == "cpu":
            return torch.device("cpu")
        if self.share_device:
            return torch.device("cuda", self.device)
        if stage == "engine":
            return torch.device("cuda", self.device + index * self.engine_tp + tp_rank)
        return torch.device("cuda", self.device + self.n_engines * self.engine_tp + (index if stage == "learner" else 0))

    @property
    def weight_group_size(self) -> int:
        """Trainer rank 0 + every inference-worker GPU (world.py:192)."""
        return 1 + self.n_engines * self.engine_tp

    def stage_names(self) -> list[str]:
        """Report names, start order: engines, learners, preprocessor, actor.  A single learner / engine keeps the bare name."""
        eng = ["engine"] if self.n_engines == 1 else [f"engine{e}" for e in range(self.n_engines)]
        lrn = ["learner"] if self.n_learners == 1 else [f"learner{r}" for r in range(self.n_learners)]
        return eng + lrn + ["preprocessor", "actor"]

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


def baseline_spec(config: int, exp_path: str, **overrides: Any) -> PipelineSpec:
    """BASELINE.json `configs[config]` as a PipelineSpec.  [1]: Qwen2.5-0.5B, one GPU, actor + learner colocated, bs 512 x seq 2048.
    [2] / [3]: Qwen2.5-7B, bs 4096 x seq 8192 on a node of 4 / 8 GPUs split by the reference's arithmetic (`world.split_gpus`, default
    fractions 4 : 0 : 4 -> 2 + 2 / 4 + 4), RCCL weight broadcast to the inference GPUs, in-flight updates.  [4]: Qwen2.5-32B, two TP = 2
    engines + 4 learners on 8 GPUs, KL-to-reference on, every TP rank receiving only its slices.  `overrides` replace fields
    (a reduced `global_batch`, `share_device=True` + `weight_transport="ipc"` to run the topology on one GPU, ...)."""
    from .world import split_gpus

    if config == 1:
        kw: dict[str, Any] = {}
    elif config in (2, 3):
        part = split_gpus({2: 4, 3: 8}[config])
        kw = dict(model="7b", global_batch=4096, seq_length=8192, n_learners=part.total_finetune_gpus, n_engines=part.total_actor_llms,
                  weight_transport="rccl", share_device=False, gradient_checkpointing=True)
        assert part.weight_update_group_size == kw["n_engines"] + 1
    elif config == 4:
        # Qwen2.5-32B, TP = 2 inference x 2 actors + 4 learner GPUs, KL-to-reference on (kl_coef: conf/deepscaler15b.yaml:34)
        part = split_gpus(8, tensor_parallel_size=2)
        kw = dict(model="32b", global_batch=4096, seq_length=8192, n_learners=part.total_finetune_gpus, n_engines=part.total_actor_llms, engine_tp=part.gpus_per_llm,
                  weight_transport="rccl", share_device=False, gradient_checkpointing=True, kl_coef=0.001)
        assert part.weight_update_group_size == 1 + kw["n_engines"] * kw["engine_tp"]
    else:
        raise ValueError("configs[1] .. [4] are the pipeline topologies ([0] is the reference's CPU plumbing case)")
    kw.update(overrides)
    return PipelineSpec(exp_path=exp_path, **kw)


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


def _hook(spec: PipelineSpec, name: str, default=None):
    """`spec.hooks` is a module name; an attribute of that module replaces the default piece (tests: a CPU policy, a torch loss)."""
    if not spec.hooks:
        return default
    import importlib

    return getattr(importlib.import_module(spec.hooks), name, default)


def _stage(fn):
    """Stage entry point: spec dict (+ the stage's index among its kind) in, a report out - an error report (with the traceback)
    when the stage dies."""

    def main(spec_dict: dict, index: int = 0) -> None:
        import sys

        spec = PipelineSpec(**spec_dict)
        for d in reversed(spec.sys_path):
            if d not in sys.path:
                sys.path.insert(0, d)
        kind = fn.__name__.replace("_stage", "")
        count = {"learner": spec.n_learners, "engine": spec.n_engines}.get(kind, 1)
        name = kind if count == 1 else f"{kind}{index}"
        logging.basicConfig(level=os.environ.get("PRL_PIPELINE_LOG", "WARNING"), format=f"%(asctime)s {name} %(levelname)s %(message)s")
        if spec.stacks_after_s:  # a stage that is still running then leaves the Python stacks of all its threads behind
            import faulthandler

            d = Path(spec.exp_path) / "reports"
            d.mkdir(parents=True, exist_ok=True)
            faulthandler.dump_traceback_later(spec.stacks_after_s, repeat=False, file=open(d / f"{name}.stacks", "w"), exit=False)
        try:
            if kind in ("learner", "engine"):
                fn(spec, index, name)
            else:
                fn(spec)
        except BaseException:  # noqa: BLE001 - the orchestrator reads it
            _report(spec, name, {"error": traceback.format_exc()})
            raise

    main.__name__ = fn.__name__
    main.__qualname__ = fn.__qualname__
    return main


def rl_config_of(spec: PipelineSpec):
    """conf/finetune/grpo.yaml over base.yaml:100-114."""
    from .finetune.rl import RLConfig

    return RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=spec.kl_coef, final_kl_coef=spec.kl_coef, clamp_log_ratio_ref_new_value=5,
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
    # one scripted llm per engine: the harness sends a group to the least busy one (actor.py:247-262)
    llms = [SyntheticLLM(spec.shape["vocab"], spec.seq_length, dense=spec.dense, prompt_max=min(512, max(8, spec.seq_length // 4)),
                         prompt_min=min(64, max(2, spec.seq_length // 32))) for _ in range(spec.n_engines)]
    harness = ActorHarness(cfg, llms, spec.exp_path, trainer_state=state, scheduler_name="actor0", wire="ragged", shuffle_seed=spec.seed)
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
                            "groups_per_model_version": {str(k): v for k, v in sorted(hist.items())}, "llm_calls": sum(l.calls for l in llms),
                            "llm_calls_per_engine": [l.calls for l in llms],
                            "pacing": {"max_lag_samples": spec.lag, "weight_update_interval": spec.weight_update_interval,
                                       "budget": ActorHarness.submission_budget(spec.attempts, 1, spec.global_batch, spec.weight_update_interval, spec.lag)}})


@_stage
def preprocessor_stage(spec: PipelineSpec) -> None:
    import torch

    from .preprocess import PreprocessorConfig, PreprocessorLoop
    from .state import TrainerState

    _set_backend(spec)
    dev = spec.device_of("preprocessor")
    torch.cuda.set_device(dev)
    state = TrainerState(Path(spec.exp_path))
    state.start_listening()
    state.wait_for_processed_samples()  # the trainer's first message (finetune_loop.py:462-465)
    # N lead trainers: `training_data/0/{rank}`, per-step quota global_batch / N each, sentinels for the ranks that are full (preprocess.py:462-481, 596-662)
    cfg = PreprocessorConfig(exp_path=Path(spec.exp_path), num_trainers=spec.n_learners, train_batch_size=1, gradient_accumulation_passes=spec.global_batch,
                             seq_length=spec.budget, attempts=spec.attempts, rl=rl_config_of(spec), eos_token_id=2, chunk_n_groups=spec.chunk_n_groups,
                             max_lag=spec.lag, samples_target=spec.steps * spec.global_batch,
                             ring_buffer_size=max(128, 2 * spec.global_batch), max_ready_samples_per_lead=max(64, spec.global_batch))
    ref_model = None
    if spec.kl_coef > 0:
        # the reference policy = the policy the run starts from, frozen, on the preprocessor's GPU: its log-probs replace the reference's
        # HTTP round trips to a second inference server (preprocess.py:86-104, llm.py:606-648)
        ref_model = build_policy(spec, dev, seed=spec.seed if spec.ref_seed is None else spec.ref_seed).eval()
        for p in ref_model.parameters():
            p.requires_grad_(False)
    loop = PreprocessorLoop(cfg, dev, trainer_state=state, profile=True, wire=spec.wire, ref_model=ref_model)
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
def engine_stage(spec: PipelineSpec, index: int = 0, name: str = "engine") -> None:
    import torch

    from .engine_update import InflightUpdateManager, ScriptedEngine, UpdateServer
    from .state import TrainerState
    from .vllm_worker import StandaloneWeightReceiver

    _set_backend(spec)
    dev = spec.device_of("engine", index)
    cuda = dev.type == "cuda"
    if cuda:
        torch.cuda.set_device(dev)
    tp = spec.engine_tp
    if tp > 1:
        # a tensor-parallel engine: `tp` workers, each holding ITS slices of every parameter in vLLM's stacked layout (qkv_proj, gate_up_proj);
        # they start at zero, so an update must really land.  Shapes / dtypes come from the policy class on the meta device: no weights here.
        from .vllm_worker import StackedShardReceiver

        meta = _hook(spec, "build_policy", build_policy)(spec, torch.device("meta"), seed=spec.seed)
        named = [(n, tuple(p.shape)) for n, p in meta.named_parameters()]
        dtypes = {n: p.dtype for n, p in meta.named_parameters()}
        devs = [spec.device_of("engine", index, t) for t in range(tp)]
        workers = [StackedShardReceiver(named, dtypes.__getitem__, devs[t], t, tp, kv_heads=spec.shape["kv"]) for t in range(tp)]
        model, worker = None, workers[0]
        engine = ScriptedEngine(workers, None)
    else:
        _trace(spec, name, "building the policy")
        model = _hook(spec, "build_policy", build_policy)(spec, dev, seed=spec.seed + 999 + index)  # different values than the trainer's: an update must really land
        _trace(spec, name, "policy built")
        model.eval()
        for p in model.parameters():
            p.requires_grad_(False)
        worker = StandaloneWeightReceiver(model, dev)
        workers = [worker]
        gen_tokens = min(256, spec.seq_length)
        ids = torch.randint(3, spec.shape["vocab"], (4, gen_tokens), device=dev)

        def generate_step():
            with torch.no_grad():
                (model.model if hasattr(model, "model") else model)(input_ids=ids)
            if cuda:
                torch.cuda.synchronize(dev)

        engine = ScriptedEngine(workers, generate_step if spec.engine_load else None)
    manager = InflightUpdateManager(engine)
    server = UpdateServer(manager)
    (Path(spec.exp_path) / "reports").mkdir(parents=True, exist_ok=True)
    tmp = Path(spec.exp_path) / "reports" / f".engine_url_{index}.tmp"
    tmp.write_text(server.url)
    tmp.rename(Path(spec.exp_path) / "reports" / f"engine_url_{index}.txt")
    _trace(spec, name, "serving " + server.url)
    if spec.weight_transport != "ipc":
        # the weight-update group: trainer rank 0 + every engine GPU, worker (engine e, TP rank t) is rank 1 + e * tp + t (vllm1.py:64-108,
        # world.py:192); with tp > 1 a worker joins the group of ITS TP rank only.  Blocks until the trainer and the other members have joined.
        import threading

        init = f"tcp://127.0.0.1:{spec.wsync_port}"
        errors: list = []

        def join(w):
            try:
                if cuda:
                    torch.cuda.set_device(w.device)
                w.init_actor_update_group(index, tp, init, spec.weight_group_size, tp_sharded=tp > 1, backend=spec.weight_transport)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=join, args=(w,)) for w in workers]
        for t_ in threads:
            t_.start()
        for t_ in threads:
            t_.join()
        if errors:
            raise errors[0]
        _trace(spec, name, f"joined the weight-update group as rank(s) {[w.pg_rank for w in workers]} of {spec.weight_group_size} ({spec.weight_transport})")
    state = TrainerState(Path(spec.exp_path))
    state.start_listening()
    t0 = time.perf_counter()
    state.wait_for_training_done(timeout=spec.stage_timeout_s)
    wall = time.perf_counter() - t0
    time.sleep(0.2)  # a POST that raced the TrainingDone message finishes
    if tp > 1:  # per TP rank: the fingerprint of ITS slices (trainer-side names -> views of the stacked storage)
        probe = [_param_probe(w._weight_shard_destinations().items()) for w in workers]
    else:
        probe = _param_probe(model.named_parameters())
    tm = manager.timings
    med = lambda k: sorted(t[k] for t in tm)[len(tm) // 2] if tm else None  # noqa: E731
    grp = getattr(worker, "model_update_group", None)
    _report(spec, name, {"updates": len(tm), "wall_s": wall, "last_version": tm[-1]["version"] if tm else None, "param_probe": probe,
                         "pause_ms_median": 1e3 * med("pause_s") if tm else None, "update_ms_median": 1e3 * med("update_s") if tm else None,
                         "resume_ms_median": 1e3 * med("resume_s") if tm else None, "per_update": tm,
                         "busy_s": sum(t["total_s"] for t in tm), "busy_frac": sum(t["total_s"] for t in tm) / max(wall, 1e-9),
                         "generation_quanta": engine.quanta, "generation_quanta_by_version": {str(k): v for k, v in engine.quanta_by_version.items()},
                         "engine_load": spec.engine_load, "device": str(dev), "weight_transport": spec.weight_transport,
                         "engine_tp": tp,
                         "weight_group": ({"rank": worker.pg_rank, "size": grp.comm_size()[0], "bytes_received": getattr(grp, "bytes_moved", None),
                                           "ranks": [w.pg_rank for w in workers],
                                           "bytes_received_per_tp_rank": [getattr(w.model_update_group, "bytes_moved", None) for w in workers]}
                                          if grp is not None else None)})
    engine.shutdown()
    for w in workers:
        w.close_communicator()
    server.close()


@_stage
def learner_stage(spec: PipelineSpec, rank: int = 0, name: str = "learner") -> None:
    import queue
    import threading

    import torch

    from . import streams
    from .finetune_loop import (TRAINER_TOPIC, LearnerStep, SamplesProcessed, StreamedLearnerStep, WeightUpdateManager, run_data_loader)

    _set_backend(spec)
    dev = spec.device_of("learner", rank)
    cuda = dev.type == "cuda"
    world, main = spec.n_learners, rank == 0
    if cuda:
        torch.cuda.set_device(dev)
    t_init = time.perf_counter()
    if world > 1:
        # the learners' own group (gradients, sample accounting, the barrier that ends an update); the weight-update group is separate
        import torch.distributed as dist

        backend = spec.learner_backend
        _trace(spec, name, f"joining the learner group: rank {rank} of {world} over {backend}")
        dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{spec.learner_port}", rank=rank, world_size=world,
                                **({"device_id": dev} if backend == "nccl" else {}))
    _trace(spec, name, "building the policy")
    model = _hook(spec, "build_policy", build_policy)(spec, dev, seed=spec.seed)  # the same seed on every rank: replicas start equal
    _trace(spec, name, "policy built")
    rl_step_fn = _hook(spec, "rl_step_fn")
    if rl_step_fn is None:
        from .fused_head import install_fused_head, rl_step_fused_head

        install_fused_head(model)
        rl_step_fn = rl_step_fused_head
    elif spec.learner == "streamed":
        raise ValueError("a hooked loss function drives LearnerStep: learner='dropin'")
    if spec.gradient_checkpointing:
        model.gradient_checkpointing_enable()
    model.train()
    n_params = sum(p.numel() for p in model.parameters())
    param_bytes = sum(p.numel() * p.element_size() for p in model.parameters())
    topic = streams.SingleStreamSpec(exp_path=Path(spec.exp_path), topic=TRAINER_TOPIC)
    urls: list[str] = []
    group = groups = None
    if main:
        # wait for every inference server like the reference does (finetune_loop.py:470)
        deadline = time.time() + spec.stage_timeout_s
        for e in range(spec.n_engines):
            url_file = Path(spec.exp_path) / "reports" / f"engine_url_{e}.txt"
            while not url_file.exists():
                if time.time() > deadline:
                    raise TimeoutError(f"engine {e} never announced its url")
                time.sleep(0.05)
            urls.append(url_file.read_text().strip())
        _trace(spec, name, "engines found at " + ", ".join(urls))
        if spec.weight_transport != "ipc" and spec.engine_tp > 1:
            from .weight_sync import weight_sync_tp_groups

            # one group per tensor-parallel rank: the trainer + that TP rank of every engine; each carries only that rank's slices
            groups = weight_sync_tp_groups(spec.weight_transport, f"tcp://127.0.0.1:{spec.wsync_port}", 0, spec.weight_group_size, spec.engine_tp, dev,
                                           timeout_s=spec.stage_timeout_s)
            group = groups[0]
            _trace(spec, name, f"{len(groups)} weight-update groups (one per TP rank) of {group.comm_size()[0]} formed over {spec.weight_transport}")
        elif spec.weight_transport != "ipc":
            from .weight_sync import weight_sync_group

            group = weight_sync_group(spec.weight_transport, f"tcp://127.0.0.1:{spec.wsync_port}", 0, spec.weight_group_size, dev, timeout_s=spec.stage_timeout_s)
            _trace(spec, name, f"weight-update group of {group.comm_size()[0]} formed over {spec.weight_transport}")
    transport = "ipc" if spec.weight_transport == "ipc" else ("sharded" if spec.engine_tp > 1 else "bucketed")
    # every rank owns a manager and calls send_weight_update (the call ends in a barrier among the learners); rank 0 sends (finetune_loop.py:205-292)
    sharded = transport == "sharded"
    mgr = WeightUpdateManager(llm_urls=urls, accelerated_model=model, update_stream=topic if main else None,
                              actor_update_group=(groups if main else [None] * spec.engine_tp) if sharded else group,
                              is_main_process=main, transport=transport, bucket_bytes=int(spec.extra.get("bucket_bytes", 1 << 30)),
                              kv_heads=spec.shape["kv"] if sharded else None)
    if main and transport == "ipc":
        from .weight_sync import ColocatedSender

        mgr._sender = ColocatedSender(dev, mgr.bucket_bytes)
        mgr._sender.rehome(model.named_parameters())  # the parameters LIVE in the exported buckets: publishing an update copies nothing
        _trace(spec, name, "parameters rehomed into the exported buckets")
    train_model = model
    if world > 1:
        # (wrapped AFTER the rehoming: DDP's reducer keeps the parameters it was given)
        train_model = torch.nn.parallel.DistributedDataParallel(model, **({"device_ids": [dev.index]} if cuda and spec.learner_backend == "nccl" else {}))
    if spec.optimizer == "sgd":
        opt = torch.optim.SGD(train_model.parameters(), lr=spec.lr)
    else:
        opt = torch.optim.AdamW(train_model.parameters(), lr=spec.lr, fused=cuda)
    rl = rl_config_of(spec)
    common = dict(train_batch_size=1, gradient_accumulation_passes=spec.global_batch, max_train_steps=spec.steps, weight_update_manager=mgr,
                  weight_update_interval=spec.weight_update_interval, trainer_stream=topic, max_lag=spec.lag)
    if spec.learner == "streamed":
        step = StreamedLearnerStep(train_model, opt, rl, **common)
    else:
        step = LearnerStep(train_model, opt, rl, rl_step_fn=rl_step_fn, **common)
    assert step.samples_per_step == spec.global_batch and step.samples_per_lead_per_step == spec.global_batch // world
    init_s = time.perf_counter() - t_init

    capture = (Path(spec.capture_step0) if world == 1 else Path(spec.capture_step0) / f"rank{rank}") if spec.capture_step0 else None
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
    _trace(spec, name, "sending weight version 0")
    mgr.send_weight_update(step.metrics.samples)
    first_sync_ms = 1e3 * (time.perf_counter() - t0)
    _trace(spec, name, f"weight version 0 acknowledged after {first_sync_ms:.0f} ms")
    def fingerprints():
        """The trainer's weights as the engines must hold them: whole, and - for tensor-parallel engines - every TP rank's slices."""
        full = _param_probe(model.named_parameters())
        if spec.engine_tp == 1:
            return full, None
        from .tp_shard import plan_tp_shards, shard_view

        named = list(model.named_parameters())
        cuts = plan_tp_shards([(n, tuple(p.shape)) for n, p in named], spec.engine_tp, spec.shape["kv"])
        return full, [_param_probe((n, shard_view(p.detach(), cuts[n], t, spec.engine_tp)) for n, p in named) for t in range(spec.engine_tp)]

    probes, tp_probes = {}, {}
    probes[str(step.metrics.samples)], tp_probes[str(step.metrics.samples)] = fingerprints()

    q: queue.Queue = queue.Queue(maxsize=8)
    stop = threading.Event()
    data_spec = streams.SingleStreamSpec(exp_path=Path(spec.exp_path), topic="training_data", partition=rank)  # this lead trainer's partition
    threading.Thread(target=run_data_loader, args=(data_spec, q, dev if cuda else None, stop), kwargs={"annotate": spec.learner == "streamed"},
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
                if cuda:
                    torch.cuda.synchronize(dev)
                torch.save(captured, capture / "step0_batches.pt")
                torch.save({n: p.detach().cpu().clone() for n, p in model.named_parameters()}, capture / "params_after.pt")
                (capture / "step0_metrics.json").write_text(json.dumps(res["metrics"]))
            if cuda:
                torch.cuda.synchronize(dev)
            t_opt = time.perf_counter()
            t1 = time.perf_counter()
            sent = step.maybe_send_weights()
            if sent:
                sync_ms.append(1e3 * (time.perf_counter() - t1))
                probes[str(step.metrics.samples)], tp_probes[str(step.metrics.samples)] = fingerprints()
            now = time.perf_counter()
            step_marks.append({"step": step.metrics.completed_steps, "wall_s": now - t_step, "waiting_for_data_s": wait_step,
                               "weight_sync_ms": sync_ms[-1] if sent else None, "compute_s": t_opt - t_step - wait_step, "loss": last_loss})
            t_step, wait_step = now, 0.0
    if cuda:
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
    wire_label = {"ipc": "hip_ipc_colocated", "rccl": "rccl_xgmi", "gloo": "gloo_host_staged"}[spec.weight_transport]
    _report(spec, name, {
        "rank": rank, "world": world, "device": str(dev), "grad_backend": spec.learner_backend if world > 1 else None,
        "weight_group": ({"size": group.comm_size()[0], "bytes_sent": getattr(group, "bytes_moved", None),
                          "bytes_sent_per_tp_rank": [getattr(g, "bytes_moved", None) for g in groups] if groups else None,
                          "param_bytes": param_bytes} if group is not None else None),
        "completed_steps": step.metrics.completed_steps, "samples": step.metrics.samples, "local_samples": step.local_samples, "micro_batches": micro_batches, "tokens": tokens,
        "wall_s": wall, "waiting_for_data_s": wait_s, "busy_s": wall - wait_s, "busy_frac": (wall - wait_s) / max(wall, 1e-9),
        "init_s": init_s, "params": n_params, "param_bytes": param_bytes,
        "steady_state": {"steps": len(steady), "s_per_step": sum(m["wall_s"] for m in steady) / max(len(steady), 1),
                         "samples_per_s": spec.global_batch * len(steady) / max(sum(m["wall_s"] for m in steady), 1e-9)},
        "per_step": step_marks,
        "weight_sync": {"transport": wire_label, "engines": spec.n_engines, "first_ms": first_sync_ms, "under_load_ms": sync_ms,
                        "median_ms": sync_sorted[len(sync_sorted) // 2] if sync_sorted else None, "max_ms": sync_sorted[-1] if sync_sorted else None,
                        "what": "send_weight_update request -> engine paused, weights copied, resumed -> HTTP ack -> WeightUpdateSuccess, while the "
                                "preprocessor's kernels and the loader's copies keep running on the same GPU"},
        "batch_queue_depth": {"median": sorted(depth)[len(depth) // 2] if depth else 0, "max": max(depth) if depth else 0, "maxsize": 8},
        "lag_optimizer_steps_histogram": {str(k): v for k, v in sorted(hist.items())},
        "lag_what": "per micro-batch: (samples trained when it is consumed - model_version stamped on its oldest rollout) // samples per step",
        "samples_too_old_to_train": step.metrics.samples_too_old_to_train, "param_probes": probes, "param_probes_per_tp_rank": tp_probes, "final_loss": last_loss, "learner": spec.learner,
        "peak_memory_GB": torch.cuda.max_memory_allocated(dev) / 1e9 if cuda else None})
    # (the exported buckets are not freed here: the engine may still have them mapped - they go with the process)
    for g in (groups or ([group] if group is not None else [])):
        g.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# orchestration
# ---------------------------------------------------------------------------------------------

STAGES = {"engine": engine_stage, "learner": learner_stage, "preprocessor": preprocessor_stage, "actor": actor_stage}


def _hooked_stage(spec_dict: dict, hook: str) -> None:
    """A stage body supplied by `spec.hooks` (tests: a host preprocessor); same contract as the built-in stages."""
    import sys

    spec = PipelineSpec(**spec_dict)
    for d in reversed(spec.sys_path):
        if d not in sys.path:
            sys.path.insert(0, d)
    name = hook.replace("_stage", "")
    try:
        _hook(spec, hook)(spec)
    except BaseException:  # noqa: BLE001 - the orchestrator reads it
        _report(spec, name, {"error": traceback.format_exc()})
        raise


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_pipeline(spec: PipelineSpec, timeout_s: float | None = None) -> dict:
    """Run the stages as processes - M engines, N learner ranks, the preprocessor, the actor - until the learners have done `spec.steps`
    optimizer steps; returns the merged report."""
    import multiprocessing as mp
    import shutil
    import sys

    from . import streams

    exp = Path(spec.exp_path)
    if (exp / "reports").exists():
        shutil.rmtree(exp / "reports")
    exp.mkdir(parents=True, exist_ok=True)
    if spec.platform == "cuda" and not spec.share_device:
        import torch

        need, have = spec.device + spec.n_engines * spec.engine_tp + spec.n_learners, torch.cuda.device_count()
        if have < need:
            raise RuntimeError(f"{spec.n_engines} engine + {spec.n_learners} learner GPUs from device {spec.device} on need {need} devices, {have} visible "
                               "(share_device=True runs the same topology on one GPU, with weight_transport 'ipc' or 'gloo')")
    if not spec.learner_port:
        spec.learner_port = _free_port()
    if not spec.wsync_port:
        spec.wsync_port = _free_port()
    for d in reversed(spec.sys_path):
        if d not in sys.path:
            sys.path.insert(0, d)
    was = (streams._backend, dict(streams._backend_options))
    _set_backend(spec, owner=True)
    streams.begin_run(exp)
    ctx = mp.get_context("spawn")
    procs = {}
    t0 = time.perf_counter()
    timeout_s = timeout_s or spec.stage_timeout_s
    names = spec.stage_names()
    try:
        for name in names:
            kind = name.rstrip("0123456789")
            index = int(name[len(kind):] or 0)
            if _hook(spec, f"{kind}_stage") is not None:
                target, args = _hooked_stage, (asdict(spec), f"{kind}_stage")
            else:
                target, args = STAGES[kind], ((asdict(spec), index) if kind in ("learner", "engine") else (asdict(spec),))
            procs[name] = ctx.Process(target=target, args=args, name=f"prl-{name}", daemon=True)
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
    for name in names:
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
        for name in names:
            f = exp / "reports" / f"{name}.trace"
            if f.exists():
                lines = f.read_text().splitlines()
                t0 = float(lines[0].split()[0]) if lines else 0.0
                traces[name] = [f"+{float(x.split()[0]) - t0:.1f}s {' '.join(x.split()[1:])}" for x in lines]
        stacks = {n: (exp / "reports" / f"{n}.stacks").read_text()[-6000:] for n in names if (exp / "reports" / f"{n}.stacks").exists()
                  and (exp / "reports" / f"{n}.stacks").stat().st_size}
        out["error"] = {"failed_stage": failed, "timed_out": timed_out, "stage_errors": {n: e[-1500:] for n, e in errors.items()}, "start_up_traces": traces,
                        "stacks": stacks}
        return out
    out["summary"] = summarize(spec, reports)
    return out


def summarize(spec: PipelineSpec, r: dict) -> dict:
    """The numbers the bench line quotes: steady-state samples/s, who was busy, what waited, weight sync under load, lag.  With N
    learners the step clock and the weight sync are rank 0's (the sender); with M engines the update timings are per engine."""
    eng = [n for n in r if n.startswith("engine")]
    lrn = [n for n in r if n.startswith("learner")]
    L, P, A, E = r[lrn[0]], r["preprocessor"], r["actor"], r[eng[0]]
    steps = max(L["completed_steps"], 1)
    busy_per_step = {"actor_s": A["busy_s"] / steps, "preprocessor_s": P["busy_s"] / steps, "learner_s": max(r[n]["busy_s"] for n in lrn) / steps,
                     "engine_s": max(r[n]["busy_s"] / max(r[n]["updates"], 1) for n in eng)}
    last = str(E.get("last_version"))
    probes_agree = None
    if last in L.get("param_probes", {}):  # EVERY engine holds the trainer's weights of the last version it acknowledged
        want = L["param_probes"][last] if spec.engine_tp == 1 else L["param_probes_per_tp_rank"][last]  # tp > 1: every TP rank ITS slices
        probes_agree = all(str(r[n].get("last_version")) == last and want == r[n]["param_probe"] for n in eng)
    out = {
        "samples_per_s": L["steady_state"]["samples_per_s"], "s_per_step": L["steady_state"]["s_per_step"], "steady_state_steps": L["steady_state"]["steps"],
        "tokens_per_s": sum(r[n]["tokens"] for n in lrn) / max(L["wall_s"], 1e-9), "optimizer_steps": L["completed_steps"],
        "busy_frac": {"actor": A["busy_frac"], "preprocessor": P["busy_frac"], **{n: r[n]["busy_frac"] for n in lrn}, **{n: r[n]["busy_frac"] for n in eng}},
        "stage_busy_s_per_step": busy_per_step,
        "sum_of_stage_busy_s_per_step": sum(busy_per_step.values()),
        "overlap": {"pipelined_s_per_step": L["wall_s"] / steps, "stages_back_to_back_s_per_step": sum(busy_per_step.values()),
                    "what": "busy seconds of every stage per optimizer step, measured inside the running pipeline, added up (= the step time if the stages "
                            "ran one after the other) next to the wall time of a step with the stages overlapping"},
        "queue_depth": {"preprocessor": P["queue_depth"], "learner_batch_queue": L["batch_queue_depth"]},
        "weight_sync_under_load_ms": {"median": L["weight_sync"]["median_ms"], "max": L["weight_sync"]["max_ms"], "first": L["weight_sync"]["first_ms"],
                                       "updates": len(L["weight_sync"]["under_load_ms"]), "transport": L["weight_sync"]["transport"],
                                       "engine_pause_update_resume_ms": [E["pause_ms_median"], E["update_ms_median"], E["resume_ms_median"]]},
        "lag_optimizer_steps_histogram": L["lag_optimizer_steps_histogram"],
        "actor_blocked_by_lag_s": A["blocked_by_lag_s"], "preprocessor_backpressure_waits": P["backpressure_waits"],
        "engine_weights_equal_trainer_at_last_version": probes_agree, "final_loss": L["final_loss"], "learner_peak_memory_GB": L["peak_memory_GB"],
    }
    if len(lrn) > 1 or len(eng) > 1 or spec.engine_tp > 1:
        out["topology"] = {"learners": len(lrn), "engines": len(eng), "engine_tp": spec.engine_tp, "grad_backend": L.get("grad_backend"), "weight_transport": L["weight_sync"]["transport"],
                           "devices": {n: r[n].get("device") for n in lrn + eng},
                           "micro_batches_per_learner": {n: r[n]["micro_batches"] for n in lrn}, "samples_per_learner": {n: r[n].get("local_samples") for n in lrn},
                           "updates_per_engine": {n: r[n]["updates"] for n in eng}}
    return out
