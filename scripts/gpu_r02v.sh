#!/bin/bash
# PMC passes over the shipped forward kernel (dual-plane, staggered): MFMA utilisation, LDS, L2, HBM fetch
set -u
OUT=gpurun_out/r02v
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_${name} -o pmc -- python $R/scripts/lmhead_fwd_only.py 2 > $R/$OUT/pmc_${name}.log 2>&1; echo "pmc $name exit $?")
  f=$(find $OUT/pmc_${name} -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "lmhead_fwd_kernel" in k:
        agg["lmhead_fwd_kernel<CfgDual, 0, true>"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
}
{
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
echo "plain timing: $(python scripts/lmhead_fwd_only.py 8 2>&1 | tail -1)"
} 2>&1 | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.db" -delete
