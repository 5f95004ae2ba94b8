#!/bin/bash
set -u
OUT=gpurun_out/r02q
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== fused head tests"
timeout 1200 python -m pytest tests/test_gpu_lmhead_fused.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest_lmhead.log | cut -c1-300
echo "== fwd"; python scripts/lmhead_fwd_only.py 8 2>&1 | tail -1
echo "== bench chunk 4096 under rocprof (kernel stats)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bwd -- python $R/scripts/lmhead_fused_bench.py --iters 3 --skip-library > $R/$OUT/lmhead_bench_4096.jsonl 2> $R/$OUT/lmhead_bench_4096.err)
grep -E 'backward' $OUT/lmhead_bench_4096.jsonl | cut -c1-250
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_4096.csv && head -14 $f | cut -c1-200
echo "== bench chunk 8192"
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --skip-library --chunk-rows 8192 > $OUT/lmhead_bench_8192.jsonl 2> $OUT/lmhead_bench_8192.err
grep -E 'backward' $OUT/lmhead_bench_8192.jsonl | cut -c1-250
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.db" -delete
