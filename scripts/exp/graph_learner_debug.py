"""(Needs scripts/exp/hip_graph_learner.patch applied: the graph mode of StreamedLearnerStep was measured and NOT adopted)
Where does the HIP-graph learner go wrong at the 0.5B shape?  StreamedLearnerStep with graph buckets on ragged packed micro-batches
(budget 2048), per micro-batch: the non-finite counter and the loss, next to an eager twin fed the same batches."""
import copy
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd import _lib  # noqa: E402
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.finetune_loop import StreamedLearnerStep, annotate_host_batch  # noqa: E402
from pipelinerl_amd.fused_head import install_fused_head  # noqa: E402
from pipelinerl_amd.pipeline_run import PipelineSpec, build_policy, rl_config_of  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402

dev = torch.device("cuda", 0)
buckets = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2048").split(","))
model_name = sys.argv[2] if len(sys.argv) > 2 else "0p5b"
n_mb, steps = 12, 3
spec = PipelineSpec(exp_path="/tmp/x", model=model_name)
rl = rl_config_of(spec)
rag_h, _ = make_ragged(8, attempts=8, seq_length=2048, vocab=spec.shape["vocab"], seed=5, prompt_min=64, prompt_max=512)
prep = populate_rl_data_ragged(rag_h.to(dev), 2, rl)
lens = rag_h.seq_lengths()
mbs, cur, used = [], [], 0
for i, n in enumerate(lens):
    if cur and used + n > 2048:
        mbs.append(cur)
        cur, used = [], 0
    cur.append(i)
    used += int(n)
mbs.append(cur)
packed = pack_prepared(prep, mbs, 2)
batches = [packed[j] for j in range(len(packed))]
print("micro-batches", len(batches), "tokens", [int(b.input_ids.shape[1]) for b in batches][:16], flush=True)


def make(graph):
    m = build_policy(spec, dev, seed=1)
    install_fused_head(m)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-6)
    st = StreamedLearnerStep(m, opt, rl, train_batch_size=1, gradient_accumulation_passes=sum(len(x) for x in mbs[:n_mb]), max_train_steps=10,
                             send_weight_updates=False, graph_buckets=buckets if graph else None)
    return m, st


results = {}
for graph in (False, True):
    m, st = make(graph)
    out = []
    for s in range(steps):
        for j in range(n_mb):
            b = batches[j]
            b.model_version = st.metrics.samples
            try:
                r = st.step(b)
            except AssertionError as e:
                rows = torch.stack(st._stats_dev).cpu()
                bad = rows[:, _lib.STAT_INDEX["nonfinite_new_logprobs"]].tolist()
                print("graph" if graph else "eager", "step", s, "assert:", e, "non-finite counter per micro-batch", bad, "losses", rows[:, _lib.STAT_INDEX["loss"]].tolist(), flush=True)
                raise SystemExit(1)
            out.append(float(r["loss"]))
        print("graph" if graph else "eager", "step", s, "done: loss sum", sum(out[-n_mb:]), "replays", st.graph_replays, "eager mbs", st.eager_micro_batches,
              "captures", {k: g.captures for k, g in st._graphs.items()}, flush=True)
    results[graph] = out
    del m, st
    torch.cuda.empty_cache()
a, b = np.array(results[False]), np.array(results[True])
print("max |loss_graph - loss_eager| per micro-batch", float(np.abs(a - b).max()), "relative", float((np.abs(a - b) / np.maximum(np.abs(a), 1e-9)).max()))
