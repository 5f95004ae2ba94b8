"""prl_split_bf16 on one micro-batch of fp32 d-logits (8192 x 152064): 4 B read + 4 B written per element."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd.lm_head import split_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.randn(8192, 152064, device=dev)
ts = []
for _ in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    hi, lo = split_bf16(g, 2)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
    del hi, lo
t = sorted(ts[1:])[len(ts[1:]) // 2]
print(f"prl_split_bf16 {g.numel()} elements: {t * 1e3:.0f} us = {g.numel() * 8 / t / 1e6:.0f} GB/s ({g.numel() * 8 / t / 1e6 / 80:.1f} % of 8 TB/s)")
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    h = g.to(torch.bfloat16)
    l = (g - h.float()).to(torch.bfloat16)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
    del h, l
print(f"three-pass torch formulation: {sorted(ts[1:])[1] * 1e3:.0f} us")
