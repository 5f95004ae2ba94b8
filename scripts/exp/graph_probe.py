"""Can one micro-batch of the pipelined learner (Qwen2.5-0.5B shape body forward -> fused head + loss -> backward) be captured in a
HIP graph, and what does replaying it cost against the eager loop?  (The eager loop is launch-bound: 48 ms of host issue per ~1500-token
micro-batch, profiles/r05b_*.)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.fused_head import install_fused_head  # noqa: E402
from pipelinerl_amd.pipeline_run import PipelineSpec, build_policy, rl_config_of  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spec = PipelineSpec(exp_path="/tmp/x")
model = build_policy(spec, dev, seed=1)
install_fused_head(model, keep_logits=True)
model.train()
rl = rl_config_of(spec)
rag_h, _ = make_ragged(1, attempts=8, seq_length=B, vocab=spec.shape["vocab"], seed=5, dense=True)
prep = populate_rl_data_ragged(rag_h.to(dev), 2, rl)
packed = pack_prepared(prep, [[i] for i in range(8)], 2)
batches = [packed[j] for j in range(8)]
head = None


def one(b):
    loss, stats = model(rl_batch=b, rl_config=rl, current_step=0, max_step=10)
    loss.backward()
    return loss, stats


for b in batches[:3]:
    one(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches:
    one(b)
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per {B}-token micro-batch", flush=True)

static = batches[0]
static.model_extra["labelled_rows"] = None
lm_head = model.lm_head
h = getattr(lm_head, "_prl_fused_lm_head")
h.skip_unlabelled = False  # every row: no data-dependent shapes inside the graph
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        one(static)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss, stats = one(static)
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:600])
    raise SystemExit(0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches:
    for name, t in b.tensors():
        if name != "seq_boundaries":
            getattr(static, name).copy_(t)
    g.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per {B}-token micro-batch; loss {float(loss):.6f}", flush=True)
eager_loss, _ = one(batches[-1])
print("eager loss on the same batch", float(eager_loss))
