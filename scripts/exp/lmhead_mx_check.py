"""Mixed-precision forward of the fused head (f16 plane + fp8 residual plane on the MX instruction) against an fp64
reference on small ragged shapes and against the two-bf16-plane forward at the Qwen2.5-7B shape; both timed interleaved."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd.fused_head import FusedLmHead  # noqa: E402

dev = torch.device("cuda", 0)


def ref64(hidden, W, ids, temp):
    z = hidden[0].double() @ W.double().t() / temp
    lp = torch.log_softmax(z, -1)
    T = ids.shape[1]
    nlp = lp[torch.arange(T - 1, device=dev), ids[0, 1:]]
    ent = -(lp.exp() * lp).sum(-1)
    return nlp, ent[:-1]


for T, H, V, wdt, hs in [(130, 64, 192, torch.float32, 1.0), (300, 64, 1088, torch.float32, 30.0), (257, 128, 320, torch.float32, 1e-3),
                         (64, 192, 4160, torch.float32, 1.0), (300, 128, 1088, torch.bfloat16, 1.0), (1024, 896, 8192, torch.float32, 1.0)]:
    g = torch.Generator(device="cpu").manual_seed(T)
    hidden = (torch.randn(1, T, H, generator=g) * hs).to(torch.bfloat16).to(dev)
    W = (torch.randn(V, H, generator=g) * (2.0 / H ** 0.5) / hs).to(wdt).to(dev)
    ids = torch.randint(0, V, (1, T), generator=g).to(dev)
    want_nlp, want_ent = ref64(hidden, W, ids, 0.7)
    out = {}
    for prec in ("bf16x2", "f16_fp8"):
        head = FusedLmHead(W, backward=False, precision=prec)
        nlp, ent, lse2, _ = head.logprob_entropy(hidden, ids, 0.7)
        torch.cuda.synchronize()
        out[prec] = (float((nlp[0, 1:].double() - want_nlp).abs().max()), float((ent[0, 1:].double() - want_ent).abs().max()))
    print(json.dumps({"shape": [T, H, V], "weight": str(wdt), "hidden_scale": hs, "max_abs_err_nlp_ent": out}))

T, H, V = 8192, 3584, 152064
torch.manual_seed(0)
hidden = torch.randn(1, T, H, device=dev).to(torch.bfloat16)
W = torch.randn(V, H, device=dev) * 0.02
ids = torch.randint(3, V, (1, T), device=dev)
heads = {p: FusedLmHead(W, backward=False, precision=p) for p in ("bf16x2", "f16_fp8")}
res = {p: heads[p].logprob_entropy(hidden, ids, 1.0) for p in heads}
torch.cuda.synchronize()
a, b = res["bf16x2"], res["f16_fp8"]
print(json.dumps({"7b_shape_mx_vs_bf16x2": {"nlp_max_abs": float((a[0] - b[0]).abs().max()), "ent_max_abs": float((a[1] - b[1]).abs().max()),
                                             "nlp_mean_abs": float((a[0] - b[0]).abs().mean()), "nlp_typical": float(a[0].abs().mean())}}))
times = {p: [] for p in heads}
for _ in range(5):
    for p, head in heads.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        head.logprob_entropy(hidden, ids, 1.0)
        e1.record()
        torch.cuda.synchronize()
        times[p].append(round(e0.elapsed_time(e1), 3))
print(json.dumps({"forward_ms": times}))
