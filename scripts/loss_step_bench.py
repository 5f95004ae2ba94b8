"""K2+K3 (GRPO loss + 32 statistics + gradient) at optimizer-step scale: ONE launch over every token
of a 4096 x 8192 step (33.5 M tokens, 1.88 GB algorithmic at 56 B/token)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config, populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.finetune.types import PipelineBatchEncoding  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    groups = int(os.environ.get("GROUPS", 512))
    V, T = 152064, 8192
    cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
                   clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, batch_size=4096)
    c_cfg, _, _ = make_loss_config(cfg, 0, 10)
    rag = make_ragged(groups, attempts=8, seq_length=T, vocab=V, seed=5, dense=True)[0].to(dev)
    prep = populate_rl_data_ragged(rag, 2, cfg)
    flat = pack_prepared(prep, [[i] for i in range(rag.n_seqs)], 2).flat
    big = PipelineBatchEncoding(**{k: v.unsqueeze(0) for k, v in flat.items()}, model_version=0, is_packed=True)
    nlp = big.old_logprobs + 0.005 * torch.randn_like(big.old_logprobs)
    ent = 3 * torch.rand_like(nlp)
    c_cfg.flat_micro_batches = 1
    n = rag.n_tokens
    for grad in (True, False):
        ms = []
        for it in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            grpo_loss_from_logprobs(c_cfg, big, nlp, ent, want_grad=grad)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        ms = sorted(ms[2:])
        by = n * (56 if grad else 52)
        print(f"K2+K3 {'with' if grad else 'no  '} grad, {n} tok: median {ms[len(ms) // 2] * 1e3:7.1f} us  min {ms[0] * 1e3:7.1f} us  "
              f"{by / ms[len(ms) // 2] / 1e6:7.1f} GB/s  {by / ms[len(ms) // 2] / 8e9 * 100:5.1f}% of 8 TB/s  "
              f"(PRL_LOSS_FAST_STATS={os.environ.get('PRL_LOSS_FAST_STATS', '1')})")


if __name__ == "__main__":
    main()
