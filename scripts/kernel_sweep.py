"""Micro-benchmarks of every kernel at BASELINE shapes (one process, interleaved rounds): prints
achieved GB/s against the algorithmic byte counts of DESIGN.md.  Run on the GPU box."""

import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pipelinerl_amd import _lib  # noqa: E402
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config, populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.finetune.types import PipelineBatchEncoding  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
PEAK = 8000.0


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def report(name, nbytes, med_ms, min_ms):
    gbs = nbytes / (med_ms * 1e-3) / 1e9
    print(f"{name:58s} {med_ms * 1e3:10.1f} us (min {min_ms * 1e3:9.1f})  {gbs:8.1f} GB/s  {100 * gbs / PEAK:5.1f}% of 8 TB/s", flush=True)


def main():
    import os

    quick = "--quick" in sys.argv  # few launches: used under rocprofv3 --pmc
    global timeit
    if quick:
        _orig = timeit
        timeit = lambda fn, iters=2, warm=1: _orig(fn, iters=2, warm=1)  # noqa: E731
    V, T = 152064, 8192
    cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
                   clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, batch_size=4096)
    c_cfg, _, _ = make_loss_config(cfg, 0, 10)

    # ---- reference point: plain device copy of the same size class
    x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    med, mn = timeit(lambda: y.copy_(x))
    report("torch copy 1 GiB (read+write)", 2 * (1 << 30), med, mn)
    z = torch.empty(2 << 30, dtype=torch.uint8, device=dev).view(torch.float32)
    med, mn = timeit(lambda: z.fill_(1.0))
    report("torch fill 2 GiB (write only)", 2 << 30, med, mn)
    med, mn = timeit(lambda: z.sum())
    report("torch sum 2 GiB fp32 (read only)", 2 << 30, med, mn)
    del z

    # ---- preprocess at step scale: 512 sequences x 8192 tokens (1/8 of the 4096-sequence step)
    rag_h, _ = make_ragged(64, attempts=8, seq_length=T, vocab=V, seed=5, dense=True)
    rag = rag_h.to(dev)
    ntok = rag.n_tokens
    mbs = [[i] for i in range(rag.n_seqs)]
    prep = populate_rl_data_ragged(rag, 2, cfg)
    med, mn = timeit(lambda: populate_rl_data_ragged(rag, 2, cfg), iters=5)
    report(f"K5 seq_scan+group_adv ({rag.n_seqs} seqs, {ntok} tok) [host incl.]", ntok * 8, med, mn)
    med, mn = timeit(lambda: pack_prepared(prep, mbs, 2), iters=5)
    report(f"K6 pack_collate ({ntok} tok) [host planning incl.]", ntok * 84, med, mn)
    batches = pack_prepared(prep, mbs, 2)
    flat = batches.flat

    # ---- K2+K3 at step scale and at micro-batch scale
    big = PipelineBatchEncoding(**{k: v.unsqueeze(0) for k, v in flat.items()}, model_version=0, is_packed=True)
    nlp = big.old_logprobs + 0.02 * torch.randn_like(big.old_logprobs)
    ent = 3 * torch.rand_like(nlp)
    flat_cfg = type(c_cfg).from_buffer_copy(c_cfg)
    flat_cfg.flat_micro_batches = 1
    med, mn = timeit(lambda: grpo_loss_from_logprobs(flat_cfg, big, nlp, ent, want_grad=True))
    report(f"K2+K3 loss+stats+grad, one launch over {ntok} tok", ntok * 56, med, mn)
    med, mn = timeit(lambda: grpo_loss_from_logprobs(flat_cfg, big, nlp, ent, want_grad=False))
    report(f"K2+K3 loss+stats (no grad), one launch over {ntok} tok", ntok * 52, med, mn)
    b0 = batches[0]
    n0, e0 = nlp[:, :T].contiguous(), ent[:, :T].contiguous()
    med, mn = timeit(lambda: grpo_loss_from_logprobs(c_cfg, b0, n0, e0, want_grad=True), iters=20)
    report(f"K2+K3 per micro-batch ({T} tok)", T * 56, med, mn)

    # ---- K1 family on one micro-batch of fp32 logits
    logits = torch.empty((1, T, V), dtype=torch.float32, device=dev).normal_(0, 2)
    grad = torch.empty_like(logits)
    ids = b0.input_ids
    o_nlp = torch.empty((1, T), dtype=torch.float32, device=dev)
    o_ent = torch.empty_like(o_nlp)
    o_lse = torch.empty_like(o_nlp)
    stream = _lib.current_stream_ptr(dev)
    fwd = lambda: _lib.check(lib.prl_logprob_entropy_fwd(1, T, V, logits.data_ptr(), 0, V, ids.data_ptr(), 1.0, o_nlp.data_ptr(), o_ent.data_ptr(), o_lse.data_ptr(), stream))  # noqa: E731
    med, mn = timeit(fwd, iters=8)
    report("K1 fwd logits->(logprob, entropy)  [read V*4/token]", T * V * 4, med, mn)
    _, _, g_nlp, _ = grpo_loss_from_logprobs(c_cfg, b0, o_nlp, o_ent, want_grad=True)
    bwd = lambda: _lib.check(lib.prl_logprob_entropy_bwd(1, T, V, logits.data_ptr(), 0, V, ids.data_ptr(), 1.0, o_lse.data_ptr(), o_ent.data_ptr(), g_nlp.data_ptr(), None, None, grad.data_ptr(), stream))  # noqa: E731
    med, mn = timeit(bwd, iters=8)
    report("K1 bwd dlogits  [read+write V*4/token]", 2 * T * V * 4, med, mn)
    # on-policy old logprobs (ratio inside the clip range): every completion row needs a full pass
    b0.old_logprobs[:, 1:] = torch.where(b0.labels[:, 1:] != -100, o_nlp[:, 1:] + 0.005 * torch.randn_like(o_nlp[:, 1:]), b0.old_logprobs[:, 1:])
    b0.ref_logprobs.copy_(b0.old_logprobs)
    _, _, g_nlp, _ = grpo_loss_from_logprobs(c_cfg, b0, o_nlp, o_ent, want_grad=True)
    print("rows with non-zero gradient:", int((g_nlp != 0).sum()), "of", T)
    med, mn = timeit(bwd, iters=8)
    report("K1 bwd dlogits, on-policy rows [read+write V*4/token]", 2 * T * V * 4, med, mn)
    fused = lambda: _lib.check(lib.prl_fused_logits_loss(  # noqa: E731
        ctypes.byref(c_cfg), 1, T, V, logits.data_ptr(), 0, V, 1.0, ids.data_ptr(), b0.labels.data_ptr(), b0.old_logprobs.data_ptr(),
        b0.ref_logprobs.data_ptr(), b0.advantages.data_ptr(), b0.rewards.data_ptr(), b0.group_tokens.data_ptr(), b0.overflow.data_ptr(),
        o_nlp.data_ptr(), o_ent.data_ptr(), o_lse.data_ptr(), grad.data_ptr(), stream))
    print("advantage of sequence 0:", float(b0.advantages[0, 0]))
    for variant in (21, 6, 0, 21):  # the shapes still built (the full round-1 sweep: profiles/r01[c-f]_kernel_sweep*.txt; round 4: r04d_*) (the full round-1 sweep: profiles/r01[c-f]_kernel_sweep*.txt)
        os.environ["PRL_FUSED_VARIANT"] = str(variant)
        med, mn = timeit(fused, iters=8)
        report(f"fused K1+grad+K1' variant {variant} [algorithmic: read+write V*4/token]", 2 * T * V * 4, med, mn)
    os.environ.pop("PRL_FUSED_VARIANT", None)
    # ---- the same micro-batch with bf16 logits in and bf16 d-logits out (a bf16 lm_head): a 304 KB row,
    #      97 % of it stays on chip between the two passes (variant 23: 16 + 2 vectors per lane)
    if not quick:
        lb = logits.to(torch.bfloat16)
        gb = torch.empty_like(lb)
        fused_bf16 = lambda: _lib.check(lib.prl_fused_logits_loss(  # noqa: E731
            ctypes.byref(c_cfg), 1, T, V, lb.data_ptr(), 1, V, 1.0, ids.data_ptr(), b0.labels.data_ptr(), b0.old_logprobs.data_ptr(),
            b0.ref_logprobs.data_ptr(), b0.advantages.data_ptr(), b0.rewards.data_ptr(), b0.group_tokens.data_ptr(), b0.overflow.data_ptr(),
            o_nlp.data_ptr(), o_ent.data_ptr(), o_lse.data_ptr(), gb.data_ptr(), stream))
        fused_bf16()  # new logprobs of the bf16 logits -> make the rollouts on-policy w.r.t. THEM
        b0.old_logprobs[:, 1:] = torch.where(b0.labels[:, 1:] != -100, o_nlp[:, 1:] + 0.005 * torch.randn_like(o_nlp[:, 1:]), b0.old_logprobs[:, 1:])
        b0.ref_logprobs.copy_(b0.old_logprobs)
        fused_bf16()
        print("bf16 rows with non-zero gradient:", int((gb.float().abs().amax(dim=-1) != 0).sum()), "of", T)
        for variant in (21, 6, 4, 0, 4):
            os.environ["PRL_FUSED_VARIANT"] = str(variant)
            med, mn = timeit(fused_bf16, iters=8)
            report(f"fused bf16 logits variant {variant} [algorithmic: read+write V*2/token]", 2 * T * V * 2, med, mn)
        os.environ.pop("PRL_FUSED_VARIANT", None)
    t0 = time.perf_counter()
    print("done", time.perf_counter() - t0)


if __name__ == "__main__":
    main()
