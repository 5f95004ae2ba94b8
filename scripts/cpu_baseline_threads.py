"""Thread sweep of the CPU baseline's loss leg (oracle.rl_loss_torch.rl_step_closed_form) on the host
of the GPU box, to pick the thread count `bench.py` reports."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import preprocess as opre  # noqa: E402
from oracle import rl_loss_torch as orlt  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries  # noqa: E402

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
T, V = 1024, 152064
rag, reasons = make_ragged(1, attempts=8, seq_length=8192, vocab=V, seed=99, dense=True)
data = opre.preprocess_chunk(ragged_to_entries(rag, reasons), 2, False)
batch = opre.collate_packed([data[0]], 2, 1)
b = {k: (v[:, :T] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in batch.items()}
cfg = dict(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
           clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, batch_size=4096, temperature=1.0)
logits = (np.random.default_rng(0).standard_normal((1, T, V)) * 2).astype(np.float32)
for n in (1, 8, 16, 32, 64, 128, 256):
    if n > (os.cpu_count() or 1):
        break
    torch.set_num_threads(n)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        orlt.rl_step_closed_form(logits, b, cfg, 0, 10, True)
        ts.append(time.perf_counter() - t0)
    print(f"threads {n:4d}: {min(ts[1:]) / T * 1e6:8.1f} us/token", flush=True)
