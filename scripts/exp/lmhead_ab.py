"""Within-process A/B of fused-head backward structures (PRL_TUNE_LMHEAD_BWD bits, chunk rows) on one Qwen2.5-7B micro-batch
(T = 8192, H = 3584, V = 152 064, fp32 weight): every variant's d hidden / d W is compared with the round-2 structure
(bit pattern 1, pinned to the oracle by tests/test_gpu_lmhead_fused.py), then the variants are timed interleaved over several
rounds (HIP events), whole backward and d-hidden-only / d-W-only.  One JSON line per variant.

A third field selects the precision of forward + recompute ("bf16x2" default, "f16_fp8" = mixed precision on the MX core); a
fourth field "keep" runs the backward from the logits the forward kept (no recompute); --fwd then also times the keeping forward.

usage: python scripts/lmhead_ab.py [--variants 1:4096,0:4096,0:8192,0:8192:f16_fp8] [--rounds 3] [--tokens 8192] [--fwd]"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd import _lib  # noqa: E402
from pipelinerl_amd.fused_head import FusedLmHead  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="1:4096,0:4096,0:8192")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--tokens", type=int, default=8192)
ap.add_argument("--fwd", action="store_true")
ap.add_argument("--hidden", type=int, default=3584)
ap.add_argument("--vocab", type=int, default=152064)
ap.add_argument("--weight-dtype", default="float32", choices=["float32", "bfloat16"])
ap.add_argument("--terms", type=int, default=3, help="hidden_grad_terms of every variant's head")
args = ap.parse_args()
dev = torch.device("cuda", 0)
T, H, V = args.tokens, args.hidden, args.vocab
torch.manual_seed(0)
hidden = torch.randn(1, T, H, device=dev).to(torch.bfloat16)
W = (torch.randn(V, H, device=dev) * 0.02).to(getattr(torch, args.weight_dtype))
ids = torch.randint(3, V, (1, T), device=dev)
g_nlp = torch.randn(1, T, device=dev) * 1e-4
variants = []
for v in args.variants.split(","):
    f = v.split(":")
    variants.append((int(f[0]), int(f[1]), (f[2] or "bf16x2") if len(f) > 2 else "bf16x2", len(f) > 3 and f[3] == "keep"))
heads, fwd, kept = {}, {}, {}
for prec in sorted({v[2] for v in variants} | {"bf16x2"}):
    heads[prec] = FusedLmHead(W, precision=prec, hidden_grad_terms=args.terms)
    if any(v[2] == prec and v[3] for v in variants):
        *fwd[prec], kept[prec] = heads[prec].logprob_entropy(hidden, ids, 1.0, keep=True)
    else:
        fwd[prec] = heads[prec].logprob_entropy(hidden, ids, 1.0)  # (nlp, ent, lse2, h): each precision's own saved statistics


def run(bits, chunk, prec="bf16x2", keep=False, want_hidden=True, want_weight=True, gw=None):
    _lib.set_tuning("lmhead_bwd", bits)
    gw = gw if gw is not None else (torch.zeros(V, H, device=dev) if want_weight else None)  # d W is always fp32
    _, ent, lse2, h = fwd[prec]
    gh = heads[prec].backward_from_token_grads(h, ids, 1.0, lse2, ent, g_nlp, None, None, want_hidden=want_hidden, grad_weight=gw,
                                               grad_hidden_dtype=torch.float32, chunk_rows=chunk, overwrite_weight_grad=True,
                                               kept_logits=kept[prec] if keep else None)
    return gh, gw


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


ref_h, ref_w = run(1, 4096)
torch.cuda.synchronize()
gw_buf = torch.zeros(V, H, device=dev)
report = {}
for bits, chunk, prec, keep in variants:
    gh, gw = run(bits, chunk, prec, keep, gw=gw_buf)
    torch.cuda.synchronize()
    report[(bits, chunk, prec, keep)] = {"bits": bits, "chunk_rows": chunk, "precision": prec, "kept_logits": keep, "d_hidden_vs_round2": rel(gh, ref_h), "d_weight_vs_round2": rel(gw, ref_w),
                             "ms": [], "ms_dh_only": [], "ms_dw_only": []}
    del gh
del ref_h, ref_w


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


for _ in range(args.rounds):
    for bits, chunk, prec, keep in variants:
        r = report[(bits, chunk, prec, keep)]
        r["ms"].append(timed(lambda: run(bits, chunk, prec, keep, gw=gw_buf)))
        r["ms_dh_only"].append(timed(lambda: run(bits, chunk, prec, keep, want_weight=False)))
        r["ms_dw_only"].append(timed(lambda: run(bits, chunk, prec, keep, want_hidden=False, gw=gw_buf)))
if args.fwd:
    for prec, head in heads.items():
        f = [round(timed(lambda: head.logprob_entropy(hidden, ids, 1.0)), 3) for _ in range(5)]
        print(json.dumps({"forward_ms": f, "precision": prec}))
        if prec in kept:
            del kept[prec]
            f = [round(timed(lambda: head.logprob_entropy(hidden, ids, 1.0, keep=True)), 3) for _ in range(5)]
            print(json.dumps({"forward_keeping_logits_ms": f, "precision": prec}))
for r in report.values():
    for k in ("ms", "ms_dh_only", "ms_dw_only"):
        r[k + "_min"] = min(r[k])
        r[k] = [round(x, 3) for x in r[k]]
    print(json.dumps(r))
