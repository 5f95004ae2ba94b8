"""(Needs scripts/exp/hip_graph_learner.patch applied: the graph mode of StreamedLearnerStep was measured and NOT adopted)
Graph replay against the SAME padded computation run eagerly (they should agree to the nondeterminism of the attention backward), step 0."""
import contextlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import pipelinerl_amd.finetune_loop as fl  # noqa: E402
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.finetune_loop import StreamedLearnerStep  # noqa: E402
from pipelinerl_amd.fused_head import install_fused_head  # noqa: E402
from pipelinerl_amd.pipeline_run import PipelineSpec, build_policy, rl_config_of  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402

dev = torch.device("cuda", 0)
n_mb = 8
spec = PipelineSpec(exp_path="/tmp/x", model="0p5b")
rl = rl_config_of(spec)
rag_h, _ = make_ragged(8, attempts=8, seq_length=2048, vocab=spec.shape["vocab"], seed=5, prompt_min=64, prompt_max=512)
prep = populate_rl_data_ragged(rag_h.to(dev), 2, rl)
packed = pack_prepared(prep, [[i] for i in range(rag_h.n_seqs)], 2)
batches = [packed[j] for j in range(n_mb)]


def make(graph):
    m = build_policy(spec, dev, seed=1)
    install_fused_head(m)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-6)
    st = StreamedLearnerStep(m, opt, rl, train_batch_size=1, gradient_accumulation_passes=n_mb + 1, max_train_steps=10, send_weight_updates=False,
                             graph_buckets=(2048,) if graph else None)
    return m, st


graph_run = fl._GraphedMicroBatch.run


def run_eager(self, batch, completed_steps):
    n = int(batch.input_ids.shape[1])
    for name, fill in self._FILL.items():
        dst, src = getattr(self.static, name), getattr(batch, name)
        dst[:, :n].copy_(src)
        if n < self.tokens:
            if fill is None:
                dst[:, n:].copy_(self._arange[:, : self.tokens - n])
            else:
                dst[:, n:].fill_(fill)
    for h in self._heads():
        h.skip_unlabelled = False
    o = self.owner
    loss, stats = o.model(rl_batch=self.static, rl_config=o.rl_config, current_step=completed_steps, max_step=o.max_train_steps)
    loss.backward()
    return loss.detach(), stats


mp, sp = make(True)   # padded, eager
sp._stream = None     # ... and on the default stream: the interfering neighbour
mp2, sp2 = make(True)  # a second eager twin: how far apart are two runs of the SAME eager computation?
sp2._stream = None
mg, sg = make(True)   # padded, graph
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
ctx = torch.cuda.stream(side) if "sidestream" in sys.argv else contextlib.nullcontext()
ctx.__enter__()
for j, b in enumerate(batches):
    fl._GraphedMicroBatch.run = run_eager
    sp.step(b)
    sp2.step(b)
    fl._GraphedMicroBatch.run = graph_run
    sg.step(b)
    torch.cuda.synchronize()
    rows = []
    for (n, pp), (_, pg) in zip(mp.named_parameters(), mg.named_parameters()):
        if pp.grad is None or pg.grad is None:
            continue
        a, c = pp.grad.float(), pg.grad.float()
        rows.append((float((a - c).abs().max() / a.abs().max().clamp_min(1e-30)), n, float(a.abs().max()), float(c.abs().max())))
    twin = max((float((a_.grad.float() - c_.grad.float()).abs().max() / a_.grad.float().abs().max().clamp_min(1e-30)) for a_, c_ in zip(mp.parameters(), mp2.parameters()) if a_.grad is not None), default=0.0)
    print(f"  two eager runs of the same padded micro-batches differ by at most {twin:.3e} (relative to the tensor's largest gradient)")
    rows.sort(reverse=True, key=lambda r: (r[0] != r[0], r[0]))
    print(f"micro-batch {j} ({int(b.input_ids.shape[1])} tokens) replays {sg.graph_replays}: graph vs eager on the same padded batch, worst parameters:", flush=True)
    for r in rows[:4]:
        print(f"    {r[0]:.3e}  {r[1]}  max|g| eager {r[2]:.3e} graph {r[3]:.3e}", flush=True)
