"""d logits as bf16 planes: the fused logits pass writing (hi, lo) planes directly against the fp32-gradient pass
followed by prl_split_bf16, on one Qwen2.5-7B micro-batch (T = 8192, V = 152 064), and the library-GEMM head
backward on top of either.  Prints JSON lines."""
import ctypes
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd import _lib  # noqa: E402
from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config  # noqa: E402
from pipelinerl_amd.lm_head import SplitBf16LmHead, _head_grads  # noqa: E402

dev = torch.device("cuda", 0)
T, H, V = 8192, 3584, 152064
lib = _lib.load()
torch.manual_seed(0)
logits = torch.randn(1, T, V, device=dev) * 2
ids = torch.randint(3, V, (1, T), device=dev)
labels = ids.clone()
labels[0, :100] = -100
cfg, _, _ = make_loss_config(RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05, batch_size=4096), 0, 10)
f = lambda v: torch.full((1, T), v, device=dev)  # noqa: E731
nlp, ent, lse = f(0.0), f(0.0), f(0.0)
# old log-probs near the model's own: every labelled row carries a gradient
lib.prl_logprob_entropy_fwd(1, T, V, logits.data_ptr(), 0, V, ids.data_ptr(), 1.0, nlp.data_ptr(), ent.data_ptr(), lse.data_ptr(), _lib.current_stream_ptr(dev))
old = nlp + 0.01 * torch.randn_like(nlp)
adv, rew, gt, ovf = torch.randn(1, T, device=dev), f(1.0), f(500.0), f(0.0)
grad = torch.empty_like(logits)
planes = torch.empty(2, T, V, dtype=torch.bfloat16, device=dev)
s = _lib.current_stream_ptr(dev)


def fp32_then_split():
    _lib.check(lib.prl_fused_logits_loss(ctypes.byref(cfg), 1, T, V, logits.data_ptr(), 0, V, 1.0, ids.data_ptr(), labels.data_ptr(), old.data_ptr(), old.data_ptr(),
                                         adv.data_ptr(), rew.data_ptr(), gt.data_ptr(), ovf.data_ptr(), nlp.data_ptr(), ent.data_ptr(), lse.data_ptr(), grad.data_ptr(), s))
    _lib.check(lib.prl_split_bf16(T * V, grad.data_ptr(), planes[0].data_ptr(), planes[1].data_ptr(), s))


def fp32_only():
    _lib.check(lib.prl_fused_logits_loss(ctypes.byref(cfg), 1, T, V, logits.data_ptr(), 0, V, 1.0, ids.data_ptr(), labels.data_ptr(), old.data_ptr(), old.data_ptr(),
                                         adv.data_ptr(), rew.data_ptr(), gt.data_ptr(), ovf.data_ptr(), nlp.data_ptr(), ent.data_ptr(), lse.data_ptr(), grad.data_ptr(), s))


def planes_direct():
    _lib.check(lib.prl_fused_logits_loss_planes(ctypes.byref(cfg), 1, T, V, logits.data_ptr(), V, 1.0, ids.data_ptr(), labels.data_ptr(), old.data_ptr(), old.data_ptr(),
                                                adv.data_ptr(), rew.data_ptr(), gt.data_ptr(), ovf.data_ptr(), nlp.data_ptr(), ent.data_ptr(), lse.data_ptr(),
                                                planes[0].data_ptr(), planes[1].data_ptr(), V, s))


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


algo = T * (2 * V * 4 + 56)
for name, fn in (("fused pass, fp32 gradient (out of place)", fp32_only), ("fused pass, fp32 gradient + prl_split_bf16", fp32_then_split),
                 ("fused pass, bf16 planes directly", planes_direct)):
    ms = timeit(fn)
    print(json.dumps({"what": name, "ms": round(ms, 3), "algorithmic_TBps": round(algo / ms / 1e9, 2), "kernel": lib.prl_last_fused_kernel().decode()}), flush=True)
a = planes.clone()
fp32_then_split()
torch.cuda.synchronize()
print(json.dumps({"what": "planes identical to split(fp32 gradient)", "equal": bool(torch.equal(a.view(torch.int16), planes.view(torch.int16)))}), flush=True)

# library backward GEMMs on the planes (d hidden 3 products + d W as one stacked GEMM)
del grad, logits
W = torch.randn(V, H, device=dev) * 0.02
head = SplitBf16LmHead(W)
parts = head._split()
x2 = torch.randn(T, H, device=dev).to(torch.bfloat16)
ms = timeit(lambda: _head_grads([planes[0], planes[1]], x2, parts, 3, True), iters=3)
print(json.dumps({"what": "library backward GEMMs on the planes (d hidden + d W)", "ms": round(ms, 3)}), flush=True)
