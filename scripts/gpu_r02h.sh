#!/bin/bash
set -u
OUT=gpurun_out/r02h
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 0 16 32 48 80; do
  echo "exp=$e: $(PRL_LMHEAD_EXP=$e PRL_LMHEAD_TILE=256x256 python scripts/lmhead_fwd_only.py 8 2>&1 | tail -1)"
done | tee $OUT/fwd_ablation.txt
export PRL_LMHEAD_TILE=256x256
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/$OUT/pmc_sq1 -o pmc -- python $R/scripts/lmhead_fwd_only.py 2 > $R/$OUT/pmc_sq1.log 2>&1)
python - $(find $OUT/pmc_sq1 -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "lmhead_fwd_kernel" in r.get("Kernel_Name", ""):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: sum(v) / len(v) for c, v in agg.items()})
PY
find $OUT -name "*kernel_trace.csv" -size +1M -delete
