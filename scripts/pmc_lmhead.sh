#!/bin/bash
# rocprofv3 --pmc passes (separate runs, kernel trace only - never combined with other trace domains) over the fused head's kernels at the
# 7B shape: matrix-pipe duty, wave-cycle split, L2 hit rate.  usage: gpurun -- 'bash scripts/pmc_lmhead.sh <tag>'
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/lmhead_ab.py --variants 0:8192::keep,0:8192 --rounds 1 --fwd"
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1; echo "pass1 $?"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- $CMD > $OUT/p2.log 2>&1; echo "pass2 $?"
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/p3 -o pmc -- $CMD > $OUT/p3.log 2>&1; echo "pass3 $?"
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70]
        d = agg[k][r["Counter_Name"]]
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
lines = ["# rocprofv3 --pmc passes (three separate runs, kernel trace only) over scripts/lmhead_ab.py --variants 0:8192::keep,0:8192 --rounds 1 --fwd",
         "# 8192 x 3584 x 152 064, fp32 weight.  GRBM_GUI_ACTIVE is reported per XCD (sum / 8 = shader cycles); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /",
         "# (1024 SIMDs x shader cycles); SQ_WAIT_* / SQ_ACTIVE_INST_ANY are fractions of SQ_WAVE_CYCLES.  Profiled launches run a few % slower."]
for k, cs in sorted(agg.items()):
    avg = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    if avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0 and "dlogits_from_kept" not in k:
        continue
    sh = avg.get("GRBM_GUI_ACTIVE", 0) / 8
    wc = avg.get("SQ_WAVE_CYCLES", 0)
    hit, miss = avg.get("TCC_HIT_sum", 0), avg.get("TCC_MISS_sum", 0)
    n = len(next(iter(cs.values())))
    lines.append(f"{k:72s} shader cycles {sh / 1e6:7.2f} M  MFMA pipes busy {100 * avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * sh) if sh else 0:5.1f} %  "
                 f"L2 hit {100 * hit / (hit + miss) if hit + miss else 0:5.1f} %  waves: parked {100 * avg.get('SQ_WAIT_ANY', 0) / wc if wc else 0:5.1f} % "
                 f"issue-stalled {100 * avg.get('SQ_WAIT_INST_ANY', 0) / wc if wc else 0:5.1f} % issuing {100 * avg.get('SQ_ACTIVE_INST_ANY', 0) / wc if wc else 0:5.1f} %  ({n} launches)")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
