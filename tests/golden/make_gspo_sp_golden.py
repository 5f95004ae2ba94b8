"""Golden vectors for GSPO under sequence parallelism (row a5: per-segment sums + SP all-reduce),
produced by running the reference's own `rl_step(..., seq_parallel_group=group)` in TWO processes
(gloo) on the two `make_slices(2)` halves of one packed batch (reference rl/utils.py:106-208,
rl/__init__.py:310-352, types.py:145-180).  Each rank's slice, logits, loss, statistics and autograd
d loss / d logits are stored (tests/golden/gspo_sp2_<case>_rank<r>.npz).

    python tests/golden/make_gspo_sp_golden.py
"""

from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))

CASES = {
    "plain": dict(policy_loss="gspo", epsilon_low=0.004, epsilon_high=0.004, kl_coef=0.1, final_kl_coef=0.1, batch_size=16),
    "groupnorm_overlong": dict(policy_loss="gspo", epsilon_low=0.002, epsilon_high=0.002, kl_coef=0.0, final_kl_coef=0.0,
                               group_normalization=True, overlong_filtering=True, batch_size=16),
}


def worker(rank: int, world: int, port: int) -> None:
    import torch.distributed as dist

    import make_golden as mg
    from pipelinerl_amd.synthetic import make_entries

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref_rl, ref_data, _ = mg.import_reference()
    for ci, (name, ck) in enumerate(CASES.items()):
        torch.manual_seed(900 + ci)  # same on both ranks: identical batch and logits
        V = 64
        cfg = ref_rl.RLConfig(**ck)
        raw = make_entries(2, attempts=3, seq_length=40, vocab=V, seed=700 + ci, prompt_min=3, prompt_max=9, with_ref=True)
        data = mg.ref_preprocess(ref_rl, ref_data, raw, cfg)
        batch = ref_data.collate_packed(data, mg.Tok(mg.EOS), seq_parallel=2 * world)
        B, L = batch.input_ids.shape
        logits = (torch.randn(B, L, V) * 2.0).float()
        z = logits[:, :-1] / cfg.temperature
        nlp = torch.log_softmax(z, -1).gather(2, batch.input_ids[:, 1:, None])[..., 0]
        old = batch.old_logprobs.clone()
        old[:, 1:] = torch.where(batch.labels[:, 1:] != -100, nlp + torch.randn_like(nlp) * 0.02, old[:, 1:])
        batch.old_logprobs = old
        ref = batch.ref_logprobs.clone()
        ref[:, 1:] = torch.where(batch.labels[:, 1:] != -100, nlp + torch.randn_like(nlp) * 0.05, ref[:, 1:])
        batch.ref_logprobs = ref
        sl = batch.make_slices(world)[rank]
        lo, hi = rank * L // world, (rank + 1) * L // world
        model = mg.FakeModel(logits[:, lo:hi].clone())
        loss, stats = ref_rl.rl_step(model, sl, 0, 10, cfg, seq_parallel_group=dist.group.WORLD)
        loss.backward()
        arrays = {f"batch/{k}": v for k, v in mg.batch_to_np(sl).items()}
        arrays["logits"] = logits[:, lo:hi].numpy()
        arrays["loss"] = np.asarray(loss.item(), dtype=np.float64)
        arrays["grad_logits"] = model.logits.grad.numpy()
        arrays["stats_keys"] = np.asarray(list(stats.keys()))
        arrays["stats_values"] = np.asarray([float(v) for v in stats.values()], dtype=np.float64)
        arrays["config_json"] = np.asarray(json.dumps(ck))
        arrays["steps"] = np.asarray([0, 10])
        arrays["full_length"] = np.asarray(L)
        np.savez_compressed(HERE / f"gspo_sp2_{name}_rank{rank}.npz", **arrays)
        print(f"[rank {rank}] gspo_sp2_{name}: slice {tuple(sl.input_ids.shape)} of L={L}, loss={loss.item():.6g}, "
              f"|grad|max={model.logits.grad.abs().max().item():.4g}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main() -> None:
    import torch.multiprocessing as mp

    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)


if __name__ == "__main__":
    main()
