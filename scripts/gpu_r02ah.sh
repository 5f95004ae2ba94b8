#!/bin/bash
# HBM-side traffic of the fused logits kernel on the round-2 tree: separate FETCH_SIZE / WRITE_SIZE passes
set -u
OUT=gpurun_out/r02ah
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/scripts/kernel_sweep.py --quick > $R/$OUT/pmc_$c.log 2>&1; echo "pmc $c exit $?")
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == sys.argv[2]:
        agg[r["Kernel_Name"].split("(")[0][-90:]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if any(s in k for s in ("fused_logits", "logprob_entropy", "grpo_loss", "pack_collate")):
        print(sys.argv[2], k, "launches", len(v), "avg_KB", round(sum(v) / len(v), 1))
PY
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.db" -delete
