#!/bin/bash
# kernel stats + PMC passes (separate runs) over the fused head backward of the round-3 tree
set -u
OUT=gpurun_out/r03g
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -o st -- python $GRAFT_REPO_ROOT/scripts/lmhead_bwd_only.py 3 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/lmhead_bwd_only.py 1 > $GRAFT_REPO_ROOT/$OUT/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/r03g"
for f in glob.glob(out + "/stats/*kernel_stats.csv"):
    print(open(f).read()[:3000])
for d in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        seen=set()
        for r in csv.DictReader(open(f)):
            key=(r["Kernel_Name"][:60], r["Dispatch_Id"])
            if key not in seen: seen.add(key); n[r["Kernel_Name"][:60]] += 1
        print("==", os.path.basename(d))
        for k, v in agg.items():
            if "gemm" in k or "lmhead" in k or "splitk" in k:
                print("  ", k, "dispatches", n[k], {c: round(x / max(n[k],1), 1) for c, x in v.items()})
PY
find $OUT -name "*kernel_trace.csv" -delete
