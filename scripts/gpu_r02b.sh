#!/bin/bash
# Round 2, GPU session B: fused head core v2 (32x32x16, 256-row tiles, 3-stage ring) - tests + bench
set -u
OUT=gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
echo "== fused head tests"
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_lmhead.log; tail -30 $OUT/pytest_lmhead.log | cut -c1-300
echo "== fused head bench"
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --skip-library > $OUT/lmhead_bench.jsonl 2> $OUT/lmhead_bench.err
echo "exit $?"; cat $OUT/lmhead_bench.jsonl | cut -c1-300; tail -5 $OUT/lmhead_bench.err
echo "== fullvocab variants re-check"
timeout 600 python -m pytest tests/test_gpu_fullvocab.py -q --timeout 300 -p no:cacheprovider -k "every_fused_variant" > $OUT/pytest_variants.log 2>&1
echo "exit $?"; tail -2 $OUT/pytest_variants.log
echo "== done"
