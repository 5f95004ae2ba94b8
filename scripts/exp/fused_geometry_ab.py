"""The fused logits kernel with more of a row held on chip ("a geometry that holds >= 92 % of the row"):
variant 21 = 1024 threads x (16 register + 9 LDS) vectors = 67 % of a 608 KB row (shipped) against
  31: 512 x (38 + 18) = 75 %   32: 768 x (28 + 12) = 81 %   35: 256 x (88 + 36) = 84 %   33: 256 x (96 + 36) = 89 %   34: 256 x (100 + 36) = 92 % (20 B/lane of scratch)
One micro-batch of 8192 x 152 064 fp32 logits, every completion row on-policy (full gradient pass), interleaved rounds in one process,
outputs compared with variant 21 first."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd import _lib  # noqa: E402
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config, populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
T, V = 8192, 152064
cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, clamp_log_ratio_ref_new_value=5, temperature=1.0,
               divide_advantage_by_std=False, batch_size=4096)
rag_h, _ = make_ragged(1, attempts=8, seq_length=T, vocab=V, seed=5, dense=True)
prep = populate_rl_data_ragged(rag_h.to(dev), 2, cfg)
b0 = pack_prepared(prep, [[i] for i in range(8)], 2)[1]
c_cfg, _, _ = make_loss_config(cfg, 0, 10)
c_cfg.skip_unlabelled = 1
logits = torch.empty((1, T, V), dtype=torch.float32, device=dev).normal_(0, 2)
grad = torch.empty_like(logits)
o_nlp = torch.empty((1, T), dtype=torch.float32, device=dev)
o_ent, o_lse = torch.empty_like(o_nlp), torch.empty_like(o_nlp)
stream = _lib.current_stream_ptr(dev)
_lib.check(lib.prl_logprob_entropy_fwd(1, T, V, logits.data_ptr(), 0, V, b0.input_ids.data_ptr(), 1.0, o_nlp.data_ptr(), o_ent.data_ptr(), o_lse.data_ptr(), stream))
b0.old_logprobs[:, 1:] = torch.where(b0.labels[:, 1:] != -100, o_nlp[:, 1:] + 0.005 * torch.randn_like(o_nlp[:, 1:]), b0.old_logprobs[:, 1:])
b0.ref_logprobs.copy_(b0.old_logprobs)


def fused():
    _lib.check(lib.prl_fused_logits_loss(
        ctypes.byref(c_cfg), 1, T, V, logits.data_ptr(), 0, V, 1.0, b0.input_ids.data_ptr(), b0.labels.data_ptr(), b0.old_logprobs.data_ptr(),
        b0.ref_logprobs.data_ptr(), b0.advantages.data_ptr(), b0.rewards.data_ptr(), b0.group_tokens.data_ptr(), b0.overflow.data_ptr(),
        o_nlp.data_ptr(), o_ent.data_ptr(), o_lse.data_ptr(), grad.data_ptr(), stream))


def run(variant, iters=10):
    _lib.set_tuning("fused_variant", variant)
    fused()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fused()
        b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    return float(np.median(ts)), float(min(ts))


variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "21,31,32,35,33,34".split(","))]
run(21, 3)
ref = (grad.clone(), o_nlp.clone(), o_ent.clone())
labelled = int((b0.labels[:, 1:] != -100).sum())
bytes_alg = (2 * labelled + (T - 1 - labelled) + 1) * V * 4  # labelled rows read + written, the others written only
print(f"{labelled} of {T} rows carry a gradient; algorithmic bytes {bytes_alg / 1e9:.3f} GB", flush=True)
for v in variants:
    if v == 21:
        continue
    run(v, 1)
    dg = float((grad - ref[0]).abs().max())
    dn = float((o_nlp - ref[1]).abs().max())
    de = float((o_ent - ref[2]).abs().max())
    print(f"variant {v} ({lib.prl_last_fused_kernel().decode()}): max |d grad| {dg:.3e} |d logprob| {dn:.3e} |d entropy| {de:.3e}", flush=True)
for rnd in range(3):
    for v in variants:
        med, mn = run(v)
        print(f"round {rnd} variant {v}: {med * 1e3:.1f} us (min {mn * 1e3:.1f}) -> {bytes_alg / med / 1e9:.3f} TB/s = {bytes_alg / med / 1e9 / 8:.4f} of 8 TB/s", flush=True)
