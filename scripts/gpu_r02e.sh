#!/bin/bash
set -u
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
echo "== fused head tests"
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_lmhead.log; tail -12 $OUT/pytest_lmhead.log | cut -c1-300
for tile in 256x256 256 128; do
  echo "tile=$tile: $(PRL_LMHEAD_TILE=$tile python scripts/lmhead_fwd_only.py 5 2>&1 | tail -1)"
done | tee $OUT/fwd_ab.txt
echo "== bench (bwd too)"
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --chunk-rows 4096 > $OUT/lmhead_bench.jsonl 2> $OUT/lmhead_bench.err
cat $OUT/lmhead_bench.jsonl | cut -c1-250; tail -3 $OUT/lmhead_bench.err
echo "== done"
