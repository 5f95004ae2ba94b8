"""Debug aid: compare the kept logits of two PRL_LMHEAD_EXP settings element by element and print where they differ."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd.fused_head import FusedLmHead  # noqa: E402

dev = torch.device("cuda", 0)
T, H, V = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
exp = sys.argv[4]
g = torch.Generator(device=dev).manual_seed(1)
hidden = torch.empty(1, T, H, device=dev).normal_(generator=g).to(torch.bfloat16)
W = torch.empty(V, H, device=dev).normal_(0.0, 0.05, generator=g)
ids = torch.randint(0, V, (1, T), device=dev, generator=g)
head = FusedLmHead(W, backward=False, keep_logits=False)
os.environ["PRL_LMHEAD_TILE"] = "256x256"
os.environ.pop("PRL_LMHEAD_EXP", None)
ref = head.logprob_entropy(hidden, ids, 1.0, keep=True)[4].clone()
os.environ["PRL_LMHEAD_EXP"] = exp
got = head.logprob_entropy(hidden, ids, 1.0, keep=True)[4].clone()
torch.cuda.synchronize()
d = (got - ref).abs()
print("max diff", float(d.max()), "of max |logit2|", float(ref.abs().max()), "wrong entries", int((d > 1e-4).sum()), "of", d.numel())
bad = (d > 1e-4)
if bad.any():
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("token rows with errors:", rows[:40].tolist(), "...", int(rows.numel()))
    print("vocab cols with errors:", cols[:40].tolist(), "...", int(cols.numel()))
    # error relative structure: ratio got/ref at a few bad entries
    idx = bad.nonzero()[:10]
    for r, c in idx.tolist():
        print(r, c, float(ref[r, c]), float(got[r, c]))
    # does the difference equal a missing / doubled 16- or 32-deep slice of the contraction?
    r, c = idx[0].tolist()
    h = hidden[0, r].float()
    w = W[c].float()
    import math
    k2 = math.log2(math.e)
    for lo in range(0, H, 16):
        part = float((h[lo:lo + 16] * w[lo:lo + 16]).sum()) * k2
        if abs(abs(part) - abs(float(got[r, c] - ref[r, c]))) < 2e-3 * max(1.0, abs(part)):
            print("difference ~ slice", lo, lo + 16, part, float(got[r, c] - ref[r, c]))
