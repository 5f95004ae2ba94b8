#!/bin/bash
# rocprofv3 kernel trace of the default bench command, the fused kernel's duration averaged PER STEP (4096 launches each):
# what the whole-command average of `--stats` mixes.  usage: gpurun -- 'bash scripts/rocprof_per_step.sh <tag>'
set -u
TAG=${1:-perstep}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-weight-sync --no-live-pmc --skip-unlabelled-steps 0 > $OUT/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "fused_logits_loss_keep_kernel" in r["Kernel_Name"]]
d.sort()
line = json.loads([l for l in open(out + "/bench.log") if l.startswith("{")][0])
per = 4096
rows = [f"# fused logits kernel, rocprofv3 kernel trace of `bench.py --steps 5 --warmup 2`: average duration per step of {per} launches ({len(d)} launches in all)"]
for s in range(len(d) // per):
    seg = [x[1] for x in d[s * per:(s + 1) * per]]
    rows.append(f"step {s} ({'warm-up' if s < 2 else 'timed  '}): avg {sum(seg) / len(seg) / 1e3:8.1f} us  min {min(seg) / 1e3:8.1f}  max {max(seg) / 1e3:8.1f}")
allavg = sum(x[1] for x in d) / len(d) / 1e3
timed = [x[1] for x in d[2 * per:]]
rows.append(f"all launches (what --stats reports): {allavg:.1f} us; timed steps only: {sum(timed) / len(timed) / 1e3:.1f} us; HIP events in the same process: {line['roofline']['avg_us']:.1f} us")
open(out + "/per_step.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
find $OUT/prof -name "*kernel_trace.csv" -delete
