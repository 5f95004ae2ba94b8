#!/bin/bash
set -u
OUT=gpurun_out/r02r
mkdir -p $OUT
export TMPDIR=/tmp
echo "== fused head tests"
timeout 1200 python -m pytest tests/test_gpu_lmhead_fused.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest_lmhead.log | cut -c1-300
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --skip-library > $OUT/lmhead_bench.jsonl 2> $OUT/lmhead_bench.err
grep -E 'backward' $OUT/lmhead_bench.jsonl | cut -c1-250
PRL_LMHEAD_DUAL=0 timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --skip-library > $OUT/lmhead_bench_nodual.jsonl 2> $OUT/lmhead_bench_nodual.err
grep -E 'backward' $OUT/lmhead_bench_nodual.jsonl | cut -c1-250
