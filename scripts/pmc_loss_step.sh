#!/bin/bash
# SQ issue/stall counters of the K2+K3 step launch (one pass, kernel-trace only).
set -u
OUT=gpurun_out/pmc_loss
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES \
   --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT -o pmc -- python $GRAFT_REPO_ROOT/scripts/loss_step_bench.py > $GRAFT_REPO_ROOT/$OUT/run.log 2>&1; echo "exit $?")
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_loss/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "grpo_loss" in k:
            print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
