#!/bin/bash
set -u
OUT=gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o st -- python $GRAFT_REPO_ROOT/scripts/pack_bench.py > $GRAFT_REPO_ROOT/$OUT/pack.txt 2>&1
cd $GRAFT_REPO_ROOT; cat $OUT/pack.txt | tail -6
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do grep -i "pack_collate\|Name" $f | cut -c1-220; grep -i "pack_collate\|Name" $f | cut -c1-260 > $OUT/pack_kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -delete
