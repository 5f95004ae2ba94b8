"""K6 pack kernel at full step size (4096 x 8192 tokens), GPU time only (plan uploaded once)."""
import ctypes, os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd import _lib
from pipelinerl_amd.finetune.data import pack_prepared
from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged
from pipelinerl_amd.synthetic import make_ragged
dev = torch.device("cuda", 0)
rag_h, _ = make_ragged(512, attempts=8, seq_length=8192, vocab=152064, seed=5, dense=True)
rag = rag_h.to(dev)
prep = populate_rl_data_ragged(rag, 2, RLConfig(divide_advantage_by_std=False))
mbs = [[i] for i in range(rag.n_seqs)]
ntok = rag.n_tokens
for tpl, nt in (("4", "0"), ("4", "1"), ("2", "0"), ("2", "1"), ("4", "0"), ("2", "1")):
    os.environ["PRL_PACK_NT"] = nt
    os.environ["PRL_PACK_TPL"] = tpl
    ts = []
    for it in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); pk = pack_prepared(prep, mbs, 2); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    t = float(np.median(ts[1:]))
    print(f"pack {ntok} tokens TPL={tpl} NT={nt}: {t*1e3:.0f} us incl. host planning -> {ntok*84/t/1e6:.0f} GB/s ({100*ntok*84/t/1e6/8000:.1f}% of 8 TB/s)")
