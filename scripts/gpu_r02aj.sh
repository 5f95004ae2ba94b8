#!/bin/bash
# d-logits kernel after the LDS-staged epilogue: HBM-side traffic
set -u
OUT=gpurun_out/r02aj
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/scripts/lmhead_bwd_only.py 2 > $R/$OUT/pmc_$c.log 2>&1; echo "pmc $c exit $?")
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    m = re.search(r"(lmhead_dlogits_kernel|gemm_nt_kernel|splitk_reduce_kernel)<([^>]*>?[^>]*)>", r["Kernel_Name"])
    if m and r["Counter_Name"] == sys.argv[2]:
        agg[m.group(1) + "<" + m.group(2)[:50] + ">"].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[2], k, "launches", len(v), "avg_KB", round(sum(v) / len(v)))
PY
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.db" -delete
