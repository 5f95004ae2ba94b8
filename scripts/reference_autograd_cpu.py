"""Build-container measurement: the REFERENCE's own `rl_step` (autograd backward) on host cores, for the
`cpu_baseline.reference_autograd` constant bench.py quotes.  Reads /root/reference, so it only runs in the
build container (the GPU box has no reference); the result is committed under profiles/.

    python scripts/reference_autograd_cpu.py [--tokens 256] [--threads N]
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=256)
ap.add_argument("--vocab", type=int, default=152064)
ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
args = ap.parse_args()

sys.path.insert(0, "/root/reference")
om = types.ModuleType("omegaconf")
om.DictConfig = om.ListConfig = om.OmegaConf = object
sys.modules.setdefault("omegaconf", om)
from pipelinerl.finetune.rl import RLConfig, rl_step  # noqa: E402
from pipelinerl.finetune.types import PipelineBatchEncoding  # noqa: E402

torch.set_num_threads(args.threads)
T, V = args.tokens, args.vocab
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.integers(3, V, size=(1, T)))
labels = ids.clone()
labels[:, : T // 8] = -100
f = lambda: torch.from_numpy(rng.standard_normal((1, T)).astype(np.float32))  # noqa: E731
old = -f().abs() * 0.7
batch = PipelineBatchEncoding(
    input_ids=ids, attention_mask=torch.ones_like(ids), labels=labels, position_ids=torch.arange(T)[None], segment_ids=torch.zeros_like(ids),
    rewards=f(), advantages=f(), ref_logprobs=old.clone(), old_logprobs=old, group_tokens=torch.full((1, T), 5000.0),
    num_labels=torch.full((1, T), float(T - T // 8)), overflow=torch.zeros(1, T), model_version=0, is_packed=True,
    seq_boundaries=torch.tensor([0, T], dtype=torch.int32),
)
logits = torch.nn.Parameter(torch.from_numpy((rng.standard_normal((1, T, V)) * 2).astype(np.float32)))
model = lambda **kw: types.SimpleNamespace(logits=logits)  # noqa: E731
cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, clamp_log_ratio_ref_new_value=5,
               divide_advantage_by_std=False, batch_size=4096, temperature=1.0)
times = []
for it in range(3):
    logits.grad = None
    t0 = time.perf_counter()
    loss, stats = rl_step(model, batch, 0, 10, cfg)
    loss.backward()
    times.append(time.perf_counter() - t0)
t = min(times[1:])
model_name = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
print(json.dumps({
    "what": "reference rl_step forward + autograd backward on the post-model loss path (fake-logits model), host cores",
    "tokens": T, "vocab": V, "threads": args.threads, "us_per_token": 1e6 * t / T, "s_per_8192_token_sample": t / T * 8192,
    "samples_per_s_extrapolated": 1.0 / (t / T * 8192), "host": {"nproc": os.cpu_count(), "model": model_name},
    "measured_in": "build container (reads /root/reference); a constant in bench.py, not re-measured on the GPU box",
}))
