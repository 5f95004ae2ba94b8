#!/bin/bash
# Memory-side rocprofv3 --pmc passes (separate runs, kernel trace only) over the fused head's kernels at the 7B shape: HBM bytes
# fetched / written per launch and the average latency of a vector-memory read request.  usage: gpurun -- 'bash scripts/pmc_lmhead_mem.sh <tag>'
set -u
TAG=${1:-pmcmem}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/lmhead_ab.py --variants 0:8192::keep,0:8192 --rounds 1 --fwd"
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1; echo "pass1 $?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- $CMD > $OUT/p2.log 2>&1; echo "pass2 $?"
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/p3 -o pmc -- $CMD > $OUT/p3.log 2>&1; echo "pass3 $?"
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
         "# 8192 x 3584 x 152 064, fp32 weight.  FETCH_SIZE (KB, the L2's fabric-side read requests: HBM and Infinity-Cache hits alike) is DOUBLED here",
         "# (MI355X_MICROARCH.md: on gfx950 it reports half the bytes of wide coalesced reads), WRITE_SIZE in KB as reported; latency = TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum in shader clocks per L1 -> L2 read request."]
for k, cs in sorted(agg.items()):
    if not any(s in k for s in ("gemm_", "lmhead_", "dlogits_from_kept")):
        continue
    avg = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    req = avg.get("TCP_TCC_READ_REQ_sum", 0)
    lines.append(f"{k:64s} beyond-L2 read {2 * avg.get('FETCH_SIZE', 0) * 1024 / 1e9:7.2f} GB  written {avg.get('WRITE_SIZE', 0) * 1024 / 1e9:6.2f} GB  "
                 f"L2 read requests {req / 1e6:8.1f} M  mean latency {avg.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / req if req else 0:7.0f} clk  "
                 f"TCP pending-stall {avg.get('TCP_PENDING_STALL_CYCLES_sum', 0) / 1e6:8.1f} Mclk  ({n} launches)")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
