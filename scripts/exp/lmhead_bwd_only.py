"""Minimal driver for profiling: N fused-head backwards (d hidden + d W) of one Qwen2.5-7B micro-batch (T = 8192,
H = 3584, V = 152 064, fp32 weight).  Used under rocprofv3 (kernel trace / PMC passes); prints the HIP-event time."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd.fused_head import FusedLmHead  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
T, H, V = 8192, 3584, 152064
torch.manual_seed(0)
hidden = torch.randn(1, T, H, device=dev).to(torch.bfloat16)
W = torch.randn(V, H, device=dev) * 0.02
ids = torch.randint(3, V, (1, T), device=dev)
head = FusedLmHead(W)
nlp, ent, lse2, h = head.logprob_entropy(hidden, ids, 1.0)
g_nlp = torch.randn(1, T, device=dev) * 1e-4
gw = torch.zeros(V, H, device=dev)
run = lambda: head.backward_from_token_grads(h, ids, 1.0, lse2, ent, g_nlp, None, None, grad_weight=gw)  # noqa: E731
run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n_iter):
    run()
b.record()
torch.cuda.synchronize()
print(f"fused backward: {a.elapsed_time(b) / n_iter:.3f} ms per call")
