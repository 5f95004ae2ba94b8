"""K6 pack kernel at full step size (4096 x 8192 tokens): the kernel alone (HIP events around the launch), for the store order
selected by PRL_EXP_PACK_BURST (1 = twelve interleaved streams, 2 / 4 = column bursts); prints column checksums so that runs of
different orders can be compared.  usage: PRL_EXP_PACK_BURST=4 python scripts/exp/pack_burst_ab.py [dense|ragged]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402


class Timer:
    def __init__(self):
        self.ev = []

    def time(self, name):
        t = self

        class C:
            def __enter__(self):
                self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.a.record()

            def __exit__(self, *e):
                self.b.record()
                t.ev.append((self.a, self.b))

        return C()


dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
rag_h, _ = make_ragged(512, attempts=8, seq_length=8192, vocab=152064, seed=5, dense=(mode == "dense"))
prep = populate_rl_data_ragged(rag_h.to(dev), 2, RLConfig(divide_advantage_by_std=False))
lens = rag_h.seq_lengths()
if mode == "dense":
    mbs = [[i] for i in range(rag_h.n_seqs)]
else:  # first-fit into 8192-token micro-batches
    mbs, cur, used = [], [], 0
    for i, n in enumerate(lens):
        if cur and used + n > 8192:
            mbs.append(cur)
            cur, used = [], 0
        cur.append(i)
        used += int(n)
    mbs.append(cur)
ntok = int(lens.sum())
timer = Timer()
for it in range(12):
    pk = pack_prepared(prep, mbs, 2, timer=timer)
torch.cuda.synchronize()
ts = [a.elapsed_time(b) for a, b in timer.ev][2:]
t = float(np.median(ts))
sums = {k: (v.double().sum().item() if v.dtype.is_floating_point else int(v.sum().item())) for k, v in pk.flat.items()}
h = int(sum((i + 1) * (int(x) if not isinstance(x, float) else int(x * 1000)) for i, x in enumerate(sums.values())) % (1 << 61))
print(f"burst={os.environ.get('PRL_EXP_PACK_BURST', 'default')} {mode}: {ntok} tokens, {len(mbs)} micro-batches, kernel {t * 1e3:.1f} us (min {min(ts) * 1e3:.1f}) -> "
      f"{ntok * 84 / t / 1e6:.0f} GB/s = {ntok * 84 / t / 1e6 / 8000:.4f} of 8 TB/s; checksum {h}", flush=True)
