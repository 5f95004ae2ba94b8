#!/bin/bash
set -u
OUT=gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
echo "== fused head tests (interleaved loads default)"
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_lmhead.log; tail -3 $OUT/pytest_lmhead.log | cut -c1-300
for il in 1 0; do for tile in 256 128; do
  echo "interleave=$il tile=$tile: $(PRL_LMHEAD_INTERLEAVE=$il PRL_LMHEAD_TILE=$tile python scripts/lmhead_fwd_only.py 5 2>&1 | tail -1)"
done; done | tee $OUT/fwd_ab.txt
echo "== bench (bwd too)"
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --skip-library > $OUT/lmhead_bench.jsonl 2> $OUT/lmhead_bench.err
grep -E "backward|default" $OUT/lmhead_bench.jsonl | cut -c1-250
echo "== done"
