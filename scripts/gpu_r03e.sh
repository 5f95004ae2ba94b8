#!/bin/bash
set -u
OUT=gpurun_out/r03e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/lmhead_ab.py --variants 1:4096,0:8192,0:8192:f16_fp8 --rounds 3 --fwd > $OUT/ab.jsonl 2>&1
echo "ab exit $?"; cat $OUT/ab.jsonl
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o st -- python $GRAFT_REPO_ROOT/scripts/lmhead_ab.py --variants 0:8192,0:8192:f16_fp8 --rounds 2 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -16 $f; cp $f $OUT/kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -delete
