"""Fused head, loss path (`fused_head_loss` forward + backward) on one Qwen2.5-7B micro-batch (T = 8192, H = 3584,
V = 152 064) as a function of the fraction of tokens that carry a label: with `skip_unlabelled` the head runs on the
labelled rows only.  Prints JSON lines."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd.finetune.rl import RLConfig  # noqa: E402
from pipelinerl_amd.finetune.types import PipelineBatchEncoding  # noqa: E402
from pipelinerl_amd.fused_head import FusedLmHead, fused_head_loss  # noqa: E402

dev = torch.device("cuda", 0)
T, H, V = 8192, 3584, 152064
g = torch.Generator(device=dev).manual_seed(5)
hidden = torch.empty(1, T, H, device=dev).normal_(generator=g).to(torch.bfloat16)
W = torch.empty(V, H, device=dev).normal_(0.0, 0.02, generator=g)
ids = torch.randint(3, V, (1, T), device=dev, generator=g)
cfg = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.01, final_kl_coef=0.01, batch_size=4096,
               clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False)
z = torch.zeros(1, T, device=dev)


def batch(frac):
    labels = ids.clone()
    n_prompt = int(round((1.0 - frac) * T / 2))  # two packed sequences, each starting with a prompt
    half = T // 2
    labels[0, :max(n_prompt, 1)] = -100
    labels[0, half:half + max(n_prompt, 1)] = -100
    pos = torch.cat([torch.arange(half, device=dev), torch.arange(T - half, device=dev)])[None]
    old = -torch.empty(1, T, device=dev).normal_(generator=g).abs() * 0.7
    return PipelineBatchEncoding(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), position_ids=pos, old_logprobs=old,
                                 ref_logprobs=old.clone(), advantages=torch.empty(1, T, device=dev).normal_(generator=g), rewards=z.clone(),
                                 group_tokens=torch.full((1, T), 5000.0, device=dev), num_labels=torch.full((1, T), float((labels != -100).sum()), device=dev),
                                 overflow=z.clone(), model_version=0, is_packed=True)


def run(head, b, w, h):
    h.grad = None
    w.grad = None
    loss, _ = fused_head_loss(h, w, head, b, cfg, 0, 10)
    loss.backward()


for skip in (False, True):
    for frac in (1.0, 0.75, 0.5, 0.25):
        w = W.clone().requires_grad_(True)
        h = hidden.clone().requires_grad_(True)
        head = FusedLmHead(w, skip_unlabelled=skip)
        b = batch(frac)
        run(head, b, w, h)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, e in ev:
            a.record()
            run(head, b, w, h)
            e.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(e) for a, e in ev)[1]
        print(json.dumps({"what": "fused_head_loss forward + backward", "skip_unlabelled": skip, "labelled_fraction": frac, "ms": round(ms, 2)}), flush=True)
        del head, w, h
        torch.cuda.empty_cache()
