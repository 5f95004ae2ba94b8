"""BASELINE `configs[1]` as the pipeline it is: actor -> preprocessor -> learner -> engine, four processes on ONE MI355X
(pipelinerl_amd/pipeline_run.py).  Prints one JSON object (the merged stage reports + summary).

    python scripts/pipeline_cfg1.py [--steps 5] [--model 0p5b] [--global-batch 512] [--seq-length 2048] [--engine-load] [--out f.json]
"""

import argparse
import json
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="0p5b", choices=["0p5b", "7b", "tiny"])
    ap.add_argument("--global-batch", type=int, default=512)
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
    ap.add_argument("--stacks-after", type=float, default=0.0, help="diagnosis: stages still alive after this many seconds dump their Python stacks")
    ap.add_argument("--timeout", type=float, default=900.0)
    ap.add_argument("--exp-path", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    from pipelinerl_amd.pipeline_run import PipelineSpec, run_pipeline

    exp = a.exp_path or tempfile.mkdtemp(prefix="prl_pipeline_")
    spec = PipelineSpec(exp_path=exp, model=a.model, global_batch=a.global_batch, seq_length=a.seq_length, pack_budget=a.pack_budget, attempts=a.attempts, steps=a.steps,
                        max_lag=a.max_lag, weight_update_interval=a.weight_update_interval, dense=a.dense, engine_load=a.engine_load,
                        gradient_checkpointing=a.gradient_checkpointing, learner=a.learner, wire=a.wire, stage_timeout_s=a.timeout, stacks_after_s=a.stacks_after)
    res = run_pipeline(spec)
    line = json.dumps(res)
    print(line)
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(line + "\n")
    return 1 if "error" in res else 0


if __name__ == "__main__":
    sys.exit(main())
