#!/bin/bash
# PMC comparison: chained (exp 0) vs refill-per-tile (exp 128) forward
set -u
OUT=gpurun_out/r02p
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # exp name counters...
  local exp=$1; shift
  local name=$1; shift
  (cd /tmp && PRL_LMHEAD_EXP=$exp timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_${name}_$exp -o pmc -- python $R/scripts/lmhead_fwd_only.py 2 > $R/$OUT/pmc_${name}_$exp.log 2>&1; echo "pmc $name exp=$exp exit $?")
  f=$(find $OUT/pmc_${name}_$exp -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "lmhead_fwd_kernel" in k:
        agg["lmhead_fwd_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
}
for e in 0 128; do
  run $e sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
  run $e tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
  run $e fetch FETCH_SIZE
  run $e tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.db" -delete
