"""Which nodes does a captured micro-batch consist of?  Dump the HIP graph (hipGraphDebugDotPrint) and count node kinds: a MEMCPY
node whose source is pageable HOST memory re-reads that address at every replay."""
import re
import sys
from collections import Counter
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pipelinerl_amd.finetune.data import pack_prepared  # noqa: E402
from pipelinerl_amd.finetune.rl import populate_rl_data_ragged  # noqa: E402
from pipelinerl_amd.fused_head import install_fused_head  # noqa: E402
from pipelinerl_amd.pipeline_run import PipelineSpec, build_policy, rl_config_of  # noqa: E402
from pipelinerl_amd.synthetic import make_ragged  # noqa: E402

dev = torch.device("cuda", 0)
out = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05p")
out.mkdir(parents=True, exist_ok=True)
spec = PipelineSpec(exp_path="/tmp/x", model=sys.argv[2] if len(sys.argv) > 2 else "0p5b")
B = 512
model = build_policy(spec, dev, seed=1)
install_fused_head(model)
model.train()
rl = rl_config_of(spec)
rag_h, _ = make_ragged(1, attempts=8, seq_length=B, vocab=spec.shape["vocab"], seed=5, dense=True)
prep = populate_rl_data_ragged(rag_h.to(dev), 2, rl)
packed = pack_prepared(prep, [[i] for i in range(8)], 2)
static = packed[0]
h = getattr(model.lm_head, "_prl_fused_lm_head", None)


def one(b):
    loss, stats = model(rl_batch=b, rl_config=rl, current_step=0, max_step=10)
    loss.backward()
    return loss, stats


one(static)
one(static)
getattr(model.lm_head, "_prl_fused_lm_head").skip_unlabelled = False
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    loss, stats = one(static)
torch.cuda.synchronize()
dot = Path("/tmp/microbatch_graph.dot")
g.debug_dump(str(dot))
text = dot.read_text() if dot.exists() else ""
kinds = Counter(re.findall(r"(MEMCPY|MEMSET|KERNEL|HOST|EVENT|EMPTY|GRAPH)", text))
print("node kinds", dict(kinds))
for line in text.split("\n"):
    if "MEMCPY" in line or "emcpy" in line:
        print(line[:400])
print(len(text), "bytes of dot")
# the same micro-batch eagerly under the profiler: every host <-> device copy with the Python frames that issued it
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one(static)
    torch.cuda.synchronize()
n = 0
for ev in prof.events():
    name = ev.name.lower()
    if "memcpy" in name or "copy_" == name or "_to_copy" in name or "hipmemcpy" in name:
        if "dtod" in name or "device -> device" in name:
            continue
        n += 1
        if n <= 40:
            print(ev.name, "| device", ev.device_type, "| shapes", getattr(ev, "input_shapes", None), "| stack", [f for f in (ev.stack or []) if "site-packages" not in f and "dist-packages/torch" not in f][:6])
print("copy-like events", n)
