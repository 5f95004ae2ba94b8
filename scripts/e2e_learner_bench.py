"""End-to-end learner step WITH a model (BASELINE config 2: Qwen2.5-0.5B GRPO, bs 512 x seq 2048, one
MI355X): random-init HuggingFace Qwen2 (stock PyTorch-ROCm forward/backward, bf16 weights, fp32
lm_head like the reference's `apply_fp32_lm_head`), AdamW, and the drop-in `LearnerStep.step()` with
the HIP loss path.  Reports samples/s and how much of the step the loss path takes.  Not the
headline benchmark (bench.py measures the hot path itself); this shows the path inside its caller.

    python scripts/e2e_learner_bench.py [--steps 2] [--micro-batch 4] [--fused]
"""

import argparse
import json
import sys
import time
import types
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--micro-batch", type=int, default=4)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--model", default="0p5b", choices=["0p5b", "7b"], help="Qwen2.5 shape (random init)")
    ap.add_argument("--fused-head", action="store_true", help="fused head: hidden states -> loss without materialising the logits (pipelinerl_amd.fused_head)")
    ap.add_argument("--no-checkpointing", action="store_true",
                    help="keep every layer's activations instead of recomputing them in the backward (fits the 288 GB of one MI355X for 7B x 8192 tokens)")
    ap.add_argument("--out", default=None, help="also write the JSON line to this file")
    args = ap.parse_args()

    import transformers

    from pipelinerl_amd.finetune.data import pad_prepared
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged, rl_step
    from pipelinerl_amd.finetune_loop import LearnerStep
    from pipelinerl_amd.synthetic import make_ragged

    dev = torch.device("cuda", 0)
    shape = {"0p5b": dict(V=151936, hidden=896, inter=4864, layers=24, heads=14, kv=2),
             "7b": dict(V=152064, hidden=3584, inter=18944, layers=28, heads=28, kv=4)}[args.model]
    V = shape["V"]
    args.layers = args.layers or shape["layers"]
    cfg_m = transformers.Qwen2Config(vocab_size=V, hidden_size=shape["hidden"], intermediate_size=shape["inter"], num_hidden_layers=args.layers,
                                     num_attention_heads=shape["heads"], num_key_value_heads=shape["kv"], max_position_embeddings=32768,
                                     tie_word_embeddings=False, attn_implementation="sdpa")
    torch.manual_seed(0)
    with torch.device(dev):  # build directly in HBM (a 7B random init on the host takes minutes)
        torch.set_default_dtype(torch.bfloat16)
        model = transformers.Qwen2ForCausalLM(cfg_m)
        torch.set_default_dtype(torch.float32)
    model = model.to(torch.bfloat16)
    model.lm_head = model.lm_head.float()  # fp32 lm_head (reference checkpoints.py:87-103)
    if args.fused_head:
        pass  # the fp32 nn.Linear stays as the parameter holder; its forward is never called
    else:
        model.lm_head.register_forward_pre_hook(lambda m, a: (a[0].float(),))
    if not args.no_checkpointing:
        model.gradient_checkpointing_enable()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-6, fused=True)
    n_params = sum(p.numel() for p in model.parameters())

    rl = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
                  clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, fused_logits_grad=args.fused)
    bs, L, mb = args.batch_size, args.seq_len, args.micro_batch
    rag_h, _ = make_ragged(bs // 8, attempts=8, seq_length=L, vocab=V, seed=1235, dense=True)
    prep = populate_rl_data_ragged(rag_h.to(dev), 2, rl)
    batches = [pad_prepared(prep, list(range(i, i + mb))) for i in range(0, bs, mb)]
    timers = {"loss_path_ms": 0.0}

    def timed_rl_step(model_, batch, cur, mx, config, seq_parallel_group=None):
        # time the post-model part only: wrap the model call so its end can be marked
        marks = {}

        def wrapped(**kw):
            out = model_(**kw)
            marks["a"] = torch.cuda.Event(enable_timing=True)
            marks["a"].record()
            return out

        if args.fused_head:
            from pipelinerl_amd.fused_head import rl_step_fused_head

            body = model_.model

            def wrapped_body(**kw):
                out = body(**kw)
                marks["a"] = torch.cuda.Event(enable_timing=True)
                marks["a"].record()
                return out

            shim = types.SimpleNamespace(model=wrapped_body, lm_head=model_.lm_head)
            loss, stats = rl_step_fused_head(shim, batch, cur, mx, config)
        else:
            loss, stats = rl_step(wrapped, batch, cur, mx, config)
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        timers.setdefault("pairs", []).append((marks["a"], b))
        return loss, stats

    step = LearnerStep(model, opt, rl, train_batch_size=mb, gradient_accumulation_passes=bs // mb, max_train_steps=100,
                       send_weight_updates=False, rl_step_fn=timed_rl_step)

    def one_step():
        for b in batches:
            res = step.step(b)
        assert res["did_optimizer_step"]
        return res

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    timers["pairs"] = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    loss_fwd_ms = sum(a.elapsed_time(b) for a, b in timers["pairs"]) / args.steps
    line = json.dumps({
        "what": "end-to-end learner step incl. model fwd/bwd + AdamW (stock PyTorch-ROCm) + HIP loss path",
        "head": "fused (log-prob / entropy in the GEMM epilogue, hand-written backward; the training forward keeps its fp32 logits for that backward)" if args.fused_head else "fp32 nn.Linear",
        "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9,
        "model": f"Qwen2.5-{args.model} shape, random init, {n_params / 1e6:.0f}M params, {args.layers} layers, bf16 + fp32 lm_head"
                 f", {'activations kept' if args.no_checkpointing else 'grad checkpointing'}, sdpa",
        "activation_recompute": not args.no_checkpointing,
        "global_batch": bs, "seq_len": L, "micro_batch": mb, "logits_mode": "fused" if args.fused else "two_pass",
        "samples_per_s": bs / dt, "s_per_step": dt, "tokens_per_s": bs * L / dt,
        "loss_forward_path_ms_per_step": loss_fwd_ms, "loss_forward_fraction": loss_fwd_ms / 1e3 / dt,
        "loss": float(res["loss"]), "rl_metrics": {k: res["metrics"].get(k) for k in ("rl/loss", "rl/ess", "rl/num_output_tokens_sum")},
    })
    print(line)
    if args.out:
        Path(args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
