"""Full-size validation of one optimizer step of the BASELINE config (7B GRPO: 4096 sequences x 8192
tokens, V = 152 064) through size-independent properties.  Run on the GPU box:

    python scripts/validate_full_step.py [n_logit_micro_batches]

Checks
  1. K5: advantages of every group sum to ~0 (leave-one-out baseline without std division),
     num_labels == completion lengths, overflow == (not finished) for the dense synthetic set.
  2. K6 (one launch, 33.5 M tokens): checksums of input_ids / old_logprobs equal the ragged source,
     position_ids are 0..8191 in every micro-batch, labels mask exactly the prompts.
  3. fused logits kernel on the first n micro-batches (default 64): rows of d logits sum to ~0,
     masked rows are exactly 0, new_logprobs <= 0, 0 < entropy <= ln V.
  4. K2+K3: the step-level launch equals the sum of per-micro-batch launches over ALL 4096
     micro-batches (loss, token counts, min/max lanes), and its gradient equals theirs.
"""

import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs  # noqa: E402
from pipelinerl_amd.hotpath import HotPathStep  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402


def main():
    n_logit_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda", 0)
    bs, T, V, attempts = 4096, 8192, 152064, 8
    t0 = time.time()
    rag_h, _ = make_ragged(bs // attempts, attempts=attempts, seq_length=T, vocab=V, seed=1236, dense=True)
    rag = rag_h.to(dev)
    print(f"synthetic step: {rag.n_seqs} sequences, {rag.n_tokens} tokens ({time.time() - t0:.1f} s)")
    cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
                   clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, batch_size=bs)
    step = HotPathStep(cfg, eos_token_id=2)
    mbs = [[i] for i in range(rag.n_seqs)]
    batches = step.preprocess(rag, mbs)
    torch.cuda.synchronize()
    flat = batches.flat

    # 1. K5
    adv = flat["advantages"].view(bs, T)[:, 0].double().cpu().numpy()
    grp = adv.reshape(-1, attempts).sum(1)
    assert np.abs(grp).max() < 1e-5, grp
    comp = np.diff(rag_h.host_lp_off)
    assert np.array_equal(flat["num_labels"].view(bs, T)[:, 0].cpu().numpy(), comp.astype(np.float32))
    assert float(flat["overflow"].sum()) == float(T) * float((rag_h.finished == 0).sum())
    print("K5 ok: group advantages sum to 0, num_labels / overflow consistent")

    # 2. K6
    assert int(flat["input_ids"].sum()) == int(rag_h.tokens.to(torch.int64).sum())
    a, b = float(flat["old_logprobs"].double().sum()), float(rag_h.logprobs.double().sum())
    assert abs(a - b) <= 1e-9 * abs(b)
    pos = flat["position_ids"].view(bs, T)
    assert torch.equal(pos, torch.arange(T, device=dev).expand(bs, T))
    n_masked = int((flat["labels"] != -100).sum())
    assert n_masked == int(comp.sum())
    assert int(flat["segment_ids"].abs().sum()) == 0 and int(flat["attention_mask"].sum()) == bs * T
    print(f"K6 ok: {bs * T} tokens packed in one launch, checksums and ids exact, {n_masked} target tokens")

    # 3. fused logits kernel
    logits = torch.empty((1, T, V), dtype=torch.float32, device=dev).normal_(0, 2)
    grad = torch.empty_like(logits)
    for j in range(n_logit_mb):
        step.logits_backward(j, logits, grad)
        if j % 16 == 0:
            rows = grad[0].double().sum(-1)
            assert rows.abs().max().item() < 2e-3, rows.abs().max().item()
            lab = batches[j].labels[0]
            masked_rows = (lab[1:] == -100).nonzero().flatten()
            assert torch.count_nonzero(grad[0, masked_rows]).item() == 0 and torch.count_nonzero(grad[0, -1]).item() == 0
    buf = step.buffers
    nlp = buf.new_logprobs[:, : n_logit_mb * T]
    ent = buf.entropy[:, : n_logit_mb * T]
    assert (nlp <= 0).all() and (ent >= 0).all() and (ent <= np.log(V) + 1e-3).all()
    print(f"fused kernel ok on {n_logit_mb} micro-batches: d-logits rows sum to 0, masked rows exactly 0")

    # 4. K2+K3 additivity over the whole step (synthetic new_logprobs for the remaining micro-batches)
    torch.manual_seed(0)
    step.buffers.new_logprobs.copy_(flat["old_logprobs"].unsqueeze(0) + 0.02 * torch.randn(1, bs * T, device=dev))
    step.buffers.entropy.uniform_(0, 3)
    loss, stats = step.finish()
    s_all = stats.cpu().numpy()
    flat_cfg = type(step.cfg).from_buffer_copy(step.cfg)
    flat_cfg.flat_micro_batches = 1
    _, _, g_all, _ = grpo_loss_from_logprobs(flat_cfg, step.step_batch, step.buffers.new_logprobs, step.buffers.entropy, want_grad=True)
    acc = np.zeros(32)
    acc_abs = np.zeros(32)
    mx = np.full(4, -np.inf)
    mn = np.full(4, np.inf)
    gdiff = 0
    for j in range(bs):
        b = batches[j]
        sl = slice(j * T, (j + 1) * T)
        _, s, g, _ = grpo_loss_from_logprobs(step.cfg, b, step.buffers.new_logprobs[:, sl], step.buffers.entropy[:, sl], want_grad=True)
        s = s.cpu().numpy()
        acc += s
        acc_abs += np.abs(s)
        mx = np.maximum(mx, s[[4, 11, 16, 26]])
        mn = np.minimum(mn, s[[5, 12, 17, 27]])
        if j % 64 == 0:
            gdiff = max(gdiff, float((g - g_all[:, sl]).abs().max()))
    assert abs(acc[0] - s_all[0]) <= 1e-9 * abs(s_all[0]), (acc[0], s_all[0])  # the loss lane is fp64 end to end
    assert acc[1] == s_all[1] == n_masked and acc[2] == s_all[2] == bs
    for k in (3, 6, 7, 8, 10, 13, 18, 19, 20, 25):
        assert abs(acc[k] - s_all[k]) <= 1e-5 * max(1.0, acc_abs[k]), (k, acc[k], s_all[k])  # fp32 per-lane partials
    assert np.array_equal(mx, s_all[[4, 11, 16, 26]]) and np.array_equal(mn, s_all[[5, 12, 17, 27]])
    assert gdiff == 0.0
    print(f"K2+K3 ok: step launch == sum of {bs} micro-batch launches (loss {s_all[0]:.6f}), gradients identical")
    print("FULL STEP VALIDATION PASSED")


if __name__ == "__main__":
    main()
