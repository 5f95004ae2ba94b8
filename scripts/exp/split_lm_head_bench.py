"""One 7B micro-batch through the output head + this package's loss kernel + the head's backward:
fp32 `nn.Linear` semantics (the reference: checkpoints.py:87-103) vs SplitBf16LmHead (bf16 MFMA GEMMs,
fp32 accumulation).  hidden [8192, 3584] bf16, weight [152064, 3584] fp32."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd.lm_head import SplitBf16LmHead  # noqa: E402

dev = torch.device("cuda", 0)
T, H, V = 8192, 3584, 152064
torch.manual_seed(0)
x = torch.randn(1, T, H, device=dev).to(torch.bfloat16).requires_grad_(True)
w = torch.nn.Parameter(torch.randn(V, H, device=dev) * 0.02)
g = torch.randn(1, T, V, device=dev) * 1e-3


def timeit(fn, iters=3):
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
    return sorted(ts)[len(ts) // 2]


def fwd_bwd(head):
    x.grad = None
    w.grad = None
    y = head(x)
    y.backward(g)


head = SplitBf16LmHead(w)
fp32 = lambda h: torch.nn.functional.linear(h.float(), w)  # noqa: E731
head1 = SplitBf16LmHead(w, hidden_grad_terms=1)
for name, fn in (("fp32 linear", fp32), ("split-bf16 head", head), ("split, 1-term dX", head1)):
    f = timeit(lambda: fn(x.detach()))
    fb = timeit(lambda: fwd_bwd(fn))
    print(f"{name:18s}: forward {f:7.2f} ms   forward + backward (dX, dW) {fb:7.2f} ms")
y32 = fp32(x.detach())
ys = head(x.detach())
import pipelinerl_amd.lm_head as _m
print("addmm epilogue accumulation:", _m._ADDMM_OUT_DTYPE)
print(f"max |split - fp32| = {(ys - y32).abs().max().item():.3e} at |logits| <= {y32.abs().max().item():.2f}")
