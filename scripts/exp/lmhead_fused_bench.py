"""One Qwen2.5-7B micro-batch (T = 8192 tokens, H = 3584, V = 152 064, fp32 head) through the fused
output head of csrc/prl_lmhead.hip vs the library path it replaces (SplitBf16LmHead GEMMs that
materialise the [T, V] logits + this package's fused logits kernel).  Prints one JSON line per
measurement; `mfma_tflops` counts the bf16 MFMA work actually executed (every plane product),
`fp32_equiv_tflops` the 2 T V H of the fp32 GEMM it stands for.

usage: python scripts/lmhead_fused_bench.py [--tokens 8192] [--iters 5] [--skip-library] [--skip-bwd]"""
import argparse
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd import _lib  # noqa: E402
from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config  # noqa: E402
from pipelinerl_amd.finetune.types import PipelineBatchEncoding  # noqa: E402
from pipelinerl_amd.fused_head import FusedLmHead  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=8192)
ap.add_argument("--hidden", type=int, default=3584)
ap.add_argument("--vocab", type=int, default=152064)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--chunk-rows", type=int, default=4096)
ap.add_argument("--skip-library", action="store_true")
ap.add_argument("--skip-bwd", action="store_true")
ap.add_argument("--bf16-weight", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda", 0)
T, H, V = args.tokens, args.hidden, args.vocab
torch.manual_seed(0)
hidden = torch.randn(1, T, H, device=dev).to(torch.bfloat16)
W = torch.randn(V, H, device=dev) * 0.02
if args.bf16_weight:
    W = W.to(torch.bfloat16)
ids = torch.randint(3, V, (1, T), device=dev)
labels = ids.clone()
labels[:, :256] = -100
f = lambda: torch.randn(1, T, device=dev)  # noqa: E731
old = -f().abs() * 0.7
batch = PipelineBatchEncoding(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), position_ids=torch.arange(T, device=dev)[None],
                              old_logprobs=old, ref_logprobs=old.clone(), advantages=f(), rewards=f(), group_tokens=torch.full((1, T), 5000.0, device=dev),
                              num_labels=torch.full((1, T), float(T - 256), device=dev), overflow=torch.zeros(1, T, device=dev), model_version=0, is_packed=True)
cfg, _, _ = make_loss_config(RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, batch_size=4096,
                                      clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False), 0, 10)


def timeit(fn, iters=args.iters):
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
    ts.sort()
    return ts[len(ts) // 2], ts[0]


planes = 1 if args.bf16_weight else 2
gemm = 2.0 * T * V * H
out = []


def emit(**kw):
    out.append(kw)
    print(json.dumps(kw), flush=True)


head = FusedLmHead(W, backward=not args.skip_bwd, chunk_rows=args.chunk_rows)
head.refresh()
torch.cuda.synchronize()
prep_ms, _ = timeit(lambda: (head.invalidate(), head.__setattr__("_key", None), head.refresh()), iters=2)
emit(what="prepare (split + transpose, once per optimizer step)", ms=round(prep_ms, 3))
for tile in ("256x256", "256", "128"):
    os.environ["PRL_LMHEAD_TILE"] = tile
    for ns in (None, "8", "16"):
        if ns is None:
            os.environ.pop("PRL_LMHEAD_NSPLIT", None)
        else:
            os.environ["PRL_LMHEAD_NSPLIT"] = ns
        med, best = timeit(lambda: head.logprob_entropy(hidden, ids, 1.0))
        emit(what="fused forward", tile=tile, nsplit=ns or "default", ms=round(med, 3),
             best_ms=round(best, 3), mfma_tflops=round(planes * gemm / med / 1e9, 1), fp32_equiv_tflops=round(gemm / med / 1e9, 1))
os.environ.pop("PRL_LMHEAD_NSPLIT", None)
os.environ.pop("PRL_LMHEAD_TILE", None)

if not args.skip_bwd:
    nlp, ent, lse2, h = head.logprob_entropy(hidden, ids, 1.0)
    _, _, g_nlp, _ = grpo_loss_from_logprobs(cfg, batch, nlp, ent)
    gw = torch.zeros(V, H, device=dev)
    for tile in ("256x256", "256", "default"):
        os.environ["PRL_LMHEAD_TILE"] = tile
        med, best = timeit(lambda: head.backward_from_token_grads(h, ids, 1.0, lse2, ent, g_nlp, None, None, grad_weight=gw), iters=max(2, args.iters // 2))
        # recompute (planes) + d hidden (planes + 1 products) + d W (2 products)
        terms = planes + (planes + 1) + 2
        emit(what="fused backward (d hidden + d W)", tile=tile, chunk_rows=args.chunk_rows,
             ms=round(med, 3), best_ms=round(best, 3), mfma_tflops=round(terms * gemm / med / 1e9, 1), fp32_equiv_tflops=round(3 * gemm / med / 1e9, 1))
    os.environ.pop("PRL_LMHEAD_TILE", None)
    for ks in (None, "1", "4"):
        if ks is None:
            os.environ.pop("PRL_LMHEAD_KSPLIT", None)
        else:
            os.environ["PRL_LMHEAD_KSPLIT"] = ks
        med, _ = timeit(lambda: head.backward_from_token_grads(h, ids, 1.0, lse2, ent, g_nlp, None, None, want_hidden=True, grad_weight=None), iters=2)
        emit(what="fused backward, d hidden only", ksplit=ks or "default", ms=round(med, 3))
    os.environ.pop("PRL_LMHEAD_KSPLIT", None)
    head.hidden_grad_terms = 1
    gw = torch.zeros(V, H, device=dev)
    med, best = timeit(lambda: head.backward_from_token_grads(h, ids, 1.0, lse2, ent, g_nlp, None, None, grad_weight=gw), iters=2)
    emit(what="fused backward, leading-term d hidden (hidden_grad_terms=1)", ms=round(med, 3), best_ms=round(best, 3))
    head.hidden_grad_terms = 3
    del gw

if not args.skip_library and not args.bf16_weight:
    import ctypes

    from pipelinerl_amd.lm_head import SplitBf16LmHead

    w = torch.nn.Parameter(W)
    lib_head = SplitBf16LmHead(w)
    x = hidden.clone().requires_grad_(True)
    lib = _lib.load()

    def library_fwd():
        with torch.no_grad():
            logits = lib_head(x)
            nlp = torch.empty(1, T, device=dev)
            _lib.check(lib.prl_logprob_entropy_fwd(1, T, V, logits.data_ptr(), 0, V, ids.data_ptr(), 1.0, nlp.data_ptr(), nlp.data_ptr(), nlp.data_ptr(),
                                                   _lib.current_stream_ptr(dev)))

    def library_fwd_bwd():
        x.grad = None
        w.grad = None
        logits = lib_head(x)
        lg = logits.detach()
        nlp, ent, lse = (torch.empty(1, T, device=dev) for _ in range(3))
        _lib.check(lib.prl_fused_logits_loss(ctypes.byref(cfg), 1, T, V, lg.data_ptr(), 0, V, 1.0, ids.data_ptr(), labels.data_ptr(), old.data_ptr(),
                                             old.data_ptr(), batch.advantages.data_ptr(), batch.rewards.data_ptr(), batch.group_tokens.data_ptr(),
                                             batch.overflow.data_ptr(), nlp.data_ptr(), ent.data_ptr(), lse.data_ptr(), lg.data_ptr(), _lib.current_stream_ptr(dev)))
        logits.backward(lg)

    med, best = timeit(library_fwd, iters=3)
    emit(what="library forward: split-bf16 GEMM (logits written) + K1 forward", ms=round(med, 3), best_ms=round(best, 3), fp32_equiv_tflops=round(gemm / med / 1e9, 1))
    med, best = timeit(library_fwd_bwd, iters=3)
    emit(what="library forward + backward: split GEMMs + fused logits kernel (logits and d logits written)", ms=round(med, 3), best_ms=round(best, 3),
         peak_extra_bytes=2 * T * V * 4)
