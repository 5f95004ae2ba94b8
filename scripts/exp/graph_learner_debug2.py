"""(Needs scripts/exp/hip_graph_learner.patch applied: the graph mode of StreamedLearnerStep was measured and NOT adopted)
Step-0 gradients of the HIP-graph learner (several buckets) against an eager twin: which parameters differ, and from which micro-batch on."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.finetune_loop import StreamedLearnerStep  # noqa: E402
from pipelinerl_amd.fused_head import install_fused_head  # noqa: E402
from pipelinerl_amd.pipeline_run import PipelineSpec, build_policy, rl_config_of  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402

dev = torch.device("cuda", 0)
buckets = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1280,1536,1792,2048").split(","))
share = (sys.argv[2] if len(sys.argv) > 2 else "share") == "share"
n_mb = 12
spec = PipelineSpec(exp_path="/tmp/x", model="0p5b")
rl = rl_config_of(spec)
rag_h, _ = make_ragged(8, attempts=8, seq_length=2048, vocab=spec.shape["vocab"], seed=5, prompt_min=64, prompt_max=512)
prep = populate_rl_data_ragged(rag_h.to(dev), 2, rl)
mbs = [[i] for i in range(rag_h.n_seqs)]
packed = pack_prepared(prep, mbs, 2)
batches = [packed[j] for j in range(n_mb)]


def make(graph):
    m = build_policy(spec, dev, seed=1)
    install_fused_head(m, keep_logits=False if "nokeep" in sys.argv else None)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-6)
    st = StreamedLearnerStep(m, opt, rl, train_batch_size=1, gradient_accumulation_passes=n_mb + 1, max_train_steps=10, send_weight_updates=False,
                             graph_buckets=buckets if graph else None)
    return m, st


me, se = make(False)
mg, sg = make(True)
if not share:
    import pipelinerl_amd.finetune_loop as fl
    orig = fl._GraphedMicroBatch.__init__

    def no_pool(self, owner, tokens, pool=None):
        orig(self, owner, tokens, pool=None)
    fl._GraphedMicroBatch.__init__ = no_pool
import contextlib  # noqa: E402

if "eagerpad" in sys.argv:  # the graph twin pads like the graph path but runs the padded micro-batch EAGERLY: calibrates what padding + full rows change by themselves
    import pipelinerl_amd.finetune_loop as fl2

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

    fl2._GraphedMicroBatch.run = run_eager

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
ctx = torch.cuda.stream(side) if "sidestream" in sys.argv else contextlib.nullcontext()
ctx.__enter__()
for j, b in enumerate(batches):
    if "noeager" not in sys.argv:
        se.step(b)
    sg.step(b)
    torch.cuda.synchronize()
    if "noeager" in sys.argv:
        bad = sum(int(not torch.isfinite(p.grad).all()) for p in mg.parameters() if p.grad is not None)
        junk = torch.randn(64 << 20, device=dev) if "junk" in sys.argv else None  # other work between replays: allocate, write, free
        del junk
        print(f"micro-batch {j}: parameters with non-finite graph gradients {bad}; replays {sg.graph_replays}", flush=True)
        continue
    worst, where, nonfinite = 0.0, None, 0
    for (n, pe), (_, pg) in zip(me.named_parameters(), mg.named_parameters()):
        if pe.grad is None or pg.grad is None:
            continue
        ge, gg = pe.grad.float(), pg.grad.float()
        if not torch.isfinite(gg).all():
            nonfinite += 1
        d = float((ge - gg).abs().max() / ge.abs().max().clamp_min(1e-30))
        if d > worst or d != d:
            worst, where = d, n
    print(f"micro-batch {j} ({int(b.input_ids.shape[1])} tokens -> bucket {next((x for x in buckets if x >= b.input_ids.shape[1]), None)}): worst relative gradient difference {worst:.3e} at {where}; "
          f"parameters with non-finite graph gradients {nonfinite}; replays {sg.graph_replays}", flush=True)
