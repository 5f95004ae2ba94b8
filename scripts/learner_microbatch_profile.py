"""Where a micro-batch of the pipelined learner spends its time (BASELINE configs[1]: Qwen2.5-0.5B shape, fused head, ragged
synthetic rollouts packed into `--budget`-token micro-batches): host wall clock per micro-batch next to the device time of its
phases (HIP events: body forward, head forward + K2/K3, head backward, body backward), for several packing budgets.

    python scripts/learner_microbatch_profile.py [--budgets 2048,4096,8192,16384] [--samples 64] [--model 0p5b]
"""

import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="0p5b")
    ap.add_argument("--budgets", default="2048,4096,8192,16384")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--seq-length", type=int, default=2048)
    ap.add_argument("--gradient-checkpointing", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    from pipelinerl_amd.finetune.rl import populate_rl_data_ragged
    from pipelinerl_amd.finetune.data import pack_prepared
    from pipelinerl_amd.fused_head import _labelled_rows, install_fused_head
    from pipelinerl_amd.hotpath import dense_micro_batches
    from pipelinerl_amd.pipeline_run import PipelineSpec, build_policy, rl_config_of
    from pipelinerl_amd.synthetic import make_ragged

    dev = torch.device("cuda", 0)
    spec = PipelineSpec(exp_path="/tmp/unused", model=a.model, seq_length=a.seq_length)
    model = build_policy(spec, dev, seed=1)
    install_fused_head(model)
    if a.gradient_checkpointing:
        model.gradient_checkpointing_enable()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-6, fused=True)
    rl = rl_config_of(spec)
    rag_h, _ = make_ragged(a.samples // 8, attempts=8, seq_length=a.seq_length, vocab=spec.shape["vocab"], seed=5)
    prep = populate_rl_data_ragged(rag_h.to(dev), 2, rl)

    # phase marks: hooks on the body and on the head's autograd function
    marks = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.setdefault(name, []).append(e)

    body = model.model
    body.register_forward_pre_hook(lambda *_: mark("body_fwd_begin"))
    body.register_forward_hook(lambda *_: mark("body_fwd_end"))
    out = []
    for budget in [int(x) for x in a.budgets.split(",")]:
        mbs = dense_micro_batches(rag_h, budget)
        packed = pack_prepared(prep, mbs, 2)
        batches = []
        for j in range(len(packed)):
            b = packed[j]
            b.model_extra["tokens"] = int(b.input_ids.numel())
            b.model_extra["labelled_rows"] = _labelled_rows(b.labels)  # what the loader thread finds on the host (no sync in the loop)
            batches.append(b)

        def run(measure: bool):
            marks.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                if measure:
                    mark("mb_begin")
                loss, stats = model(rl_batch=b, rl_config=rl, current_step=0, max_step=10)
                if measure:
                    mark("loss_ready")
                loss.backward()
                if measure:
                    mark("mb_end")
            host_s = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            opt.step()
            opt.zero_grad()
            return host_s, wall

        run(False)
        run(False)
        host_s, wall = run(True)
        n = len(batches)
        tok = sum(int(b.input_ids.numel()) for b in batches)

        def span(x, y):
            return sum(p.elapsed_time(q) for p, q in zip(marks[x], marks[y])) / n

        res = {"budget": budget, "micro_batches": n, "tokens": tok, "tokens_per_micro_batch": tok / n,
               "wall_ms_per_micro_batch": 1e3 * wall / n, "host_issue_ms_per_micro_batch": 1e3 * host_s / n, "us_per_token": 1e6 * wall / tok,
               "device_ms": {"body_forward": span("body_fwd_begin", "body_fwd_end"), "head_forward_and_loss": span("body_fwd_end", "loss_ready"),
                             "backward_head_and_body": span("loss_ready", "mb_end"), "whole": span("mb_begin", "mb_end")},
               "samples_per_s_at_this_rate": a.samples / wall, "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9}
        print(json.dumps(res), flush=True)
        out.append(res)
    if a.out:
        Path(a.out).write_text("\n".join(json.dumps(r) for r in out) + "\n")


if __name__ == "__main__":
    main()
