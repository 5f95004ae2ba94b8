"""The hot path as the pipeline it is (pipelinerl_amd/pipeline_run.py): actor -> preprocessor -> N learner ranks -> M engines as OS processes.
Prints one JSON object (the merged stage reports + summary).  Defaults = BASELINE `configs[1]` (four processes on ONE MI355X).

    python scripts/pipeline_cfg1.py [--steps 5] [--model 0p5b] [--global-batch 512] [--seq-length 2048] [--engine-load] [--out f.json]
    # configs[2] topology on ONE GPU (gloo gradients, HIP IPC to both engines):
    python scripts/pipeline_cfg1.py --learners 2 --engines 2 --weights ipc
    # configs[2] / [3] on their own GPUs (engines first, learners after them; RCCL gradients and weight-update group):
    python scripts/pipeline_cfg1.py --model 7b --global-batch 4096 --seq-length 8192 --learners 2 --engines 2 --weights rccl --own-gpus --gradient-checkpointing
"""

import argparse
import json
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[k] as given (pipeline_run.baseline_spec): [2] = 7B, 2 + 2 GPUs, RCCL; [3] = 4 + 4; other flags are ignored "
                         "except --steps / --global-batch / --timeout / --exp-path / --out")
    ap.add_argument("--model", default="0p5b", choices=["0p5b", "7b", "tiny"])
    ap.add_argument("--global-batch", type=int, default=None, help="default: 512, or the config's")
    ap.add_argument("--seq-length", type=int, default=2048)
    ap.add_argument("--pack-budget", type=int, default=None, help="tokens per packed micro-batch (default: --seq-length)")
    ap.add_argument("--attempts", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--max-lag", type=int, default=None, help="samples (default: one optimizer step)")
    ap.add_argument("--weight-update-interval", type=int, default=1)
    ap.add_argument("--dense", action="store_true", help="every rollout exactly seq_length tokens")
    ap.add_argument("--engine-load", action="store_true", help="the engine runs forward passes between updates")
    ap.add_argument("--gradient-checkpointing", action="store_true")
    ap.add_argument("--learner", default="streamed", choices=["streamed", "dropin"])
    ap.add_argument("--wire", default="full", choices=["full", "compact"], help="training_data records: the expanded batch, or the ragged columns (K6 on the learner's GPU)")
    ap.add_argument("--learners", type=int, default=1, help="data-parallel learner ranks (lead trainers)")
    ap.add_argument("--engines", type=int, default=1, help="inference engines (weight-update group = engines + 1)")
    ap.add_argument("--engine-tp", type=int, default=1, help="tensor-parallel degree of an engine (> 1: every TP rank receives only its slices; --weights rccl | gloo)")
    ap.add_argument("--weights", default="ipc", choices=["ipc", "rccl", "gloo"], help="trainer -> engines transport")
    ap.add_argument("--own-gpus", action="store_true", help="one GPU per engine and per learner rank (engines first); default: every stage on GPU 0")
    ap.add_argument("--kl-coef", type=float, default=0.0, help="> 0: KL-to-reference on, the preprocessor holds the frozen reference policy (configs[4]: 0.001)")
    ap.add_argument("--stacks-after", type=float, default=0.0, help="diagnosis: stages still alive after this many seconds dump their Python stacks")
    ap.add_argument("--timeout", type=float, default=900.0)
    ap.add_argument("--exp-path", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    exp = a.exp_path or tempfile.mkdtemp(prefix="prl_pipeline_")
    if a.config is not None:
        from pipelinerl_amd.pipeline_run import baseline_spec

        over = {"steps": a.steps, "stage_timeout_s": a.timeout, "stacks_after_s": a.stacks_after}
        if a.global_batch:
            over["global_batch"] = a.global_batch
        spec = baseline_spec(a.config, exp, **over)
    else:
        spec = _spec_from_flags(a, exp, PipelineSpec)
    res = run_pipeline(spec)
    line = json.dumps(res)
    print(line)
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(line + "\n")
    return 1 if "error" in res else 0


def _spec_from_flags(a, exp, PipelineSpec):
    return PipelineSpec(exp_path=exp, model=a.model, global_batch=a.global_batch or 512, seq_length=a.seq_length, pack_budget=a.pack_budget, attempts=a.attempts, steps=a.steps,
                        max_lag=a.max_lag, weight_update_interval=a.weight_update_interval, dense=a.dense, engine_load=a.engine_load,
                        gradient_checkpointing=a.gradient_checkpointing, learner=a.learner, wire=a.wire, stage_timeout_s=a.timeout, stacks_after_s=a.stacks_after,
                        n_learners=a.learners, n_engines=a.engines, engine_tp=a.engine_tp, weight_transport=a.weights, share_device=not a.own_gpus, kl_coef=a.kl_coef)


if __name__ == "__main__":
    sys.exit(main())
