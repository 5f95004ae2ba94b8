#!/bin/bash
set -u
OUT=gpurun_out/r02o
mkdir -p $OUT
export TMPDIR=/tmp
echo "== fused head tests"
timeout 1200 python -m pytest tests/test_gpu_lmhead_fused.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?"; tail -3 $OUT/pytest_lmhead.log | cut -c1-300
for e in 0 128 0 128; do echo "exp=$e: $(PRL_LMHEAD_EXP=$e python scripts/lmhead_fwd_only.py 8 2>&1 | tail -1)"; done | tee $OUT/fwd_ab.txt
echo "== bench"
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --skip-library > $OUT/lmhead_bench.jsonl 2> $OUT/lmhead_bench.err
grep -E '"nsplit": "default"|backward' $OUT/lmhead_bench.jsonl | cut -c1-250; tail -2 $OUT/lmhead_bench.err
