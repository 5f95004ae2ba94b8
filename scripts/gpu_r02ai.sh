#!/bin/bash
# PMC passes over the backward kernels of the fused head
set -u
OUT=gpurun_out/r02ai
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$OUT/pmc_${name} -o pmc -- python $R/scripts/lmhead_bwd_only.py 2 > $R/$OUT/pmc_${name}.log 2>&1; echo "pmc $name exit $?")
  f=$(find $OUT/pmc_${name} -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    m = re.search(r"(lmhead_dlogits_kernel|gemm_nt_kernel|splitk_reduce_kernel)<([^>]*>?[^>]*)>", k)
    if m:
        agg[m.group(1) + "<" + m.group(2)[:60] + ">"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
}
{
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
echo "plain timing: $(python scripts/lmhead_bwd_only.py 3 2>&1 | tail -1)"
} 2>&1 | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.db" -delete
