"""How exact is d hidden / d W of the fused head when it is asked for fp32 (a tied fp32 weight, V = 512, H = 64, and the 7B-like
H = 896 / V = 151936 slice)?  Relative 2-norm error against the fp64 product, per `hidden_grad_terms`, kept logits vs recompute."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd.fused_head import FusedLmHead  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for (T, H, V) in ((256, 64, 512), (1024, 896, 151936 // 8)):
    W = torch.empty(V, H, device=dev).normal_(0, 0.02, generator=g)
    h = torch.empty(1, T, H, device=dev).normal_(generator=g).to(torch.bfloat16)
    ids = torch.randint(0, V, (1, T), device=dev, generator=g)
    g_nlp = torch.empty(1, T, device=dev).normal_(generator=g) * 1e-3
    for terms in (3, 2, 1):
        for keep in (True, False):
            head = FusedLmHead(W, hidden_grad_terms=terms, keep_logits=keep)
            out = head.logprob_entropy(h, ids, 1.0, keep=keep)
            nlp, ent, lse2, hb = out[:4]
            kept = out[4] if keep else None
            gw = torch.zeros(V, H, device=dev)
            gh = head.backward_from_token_grads(hb, ids, 1.0, lse2, ent, g_nlp, None, None, grad_weight=gw, grad_hidden_dtype=torch.float32, kept_logits=kept)
            # fp64 reference: z = h W^T, nlp[u] = z[u-1, ids[u]] - lse(z[u-1]); d z[u-1] = g[u] (onehot - p)
            z = h[0].double() @ W.double().t()
            p = torch.softmax(z, dim=-1)
            dz = torch.zeros_like(z)
            gu = g_nlp[0, 1:].double()
            dz[:-1] = -p[:-1] * gu[:, None]
            dz[torch.arange(T - 1, device=dev), ids[0, 1:]] += gu
            want_h = dz @ W.double()
            want_w = dz.t() @ h[0].double()
            nlp_ref = torch.log_softmax(z, -1)[torch.arange(T - 1, device=dev), ids[0, 1:]]
            e = lambda a, b: float((a.double() - b).norm() / b.norm())  # noqa: E731
            print(f"T {T} H {H} V {V} terms {terms} keep {keep}: d hidden rel err {e(gh[0], want_h):.2e}  d W rel err {e(gw, want_w):.2e}  nlp abs err {float((nlp[0, 1:].double() - nlp_ref).abs().max()):.2e}", flush=True)
