#!/bin/bash
set -u
OUT=gpurun_out/r02l
mkdir -p $OUT
export TMPDIR=/tmp
echo "== fused head tests (dual-plane core default for 256x256)"
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py tests/test_gpu_parity.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_lmhead.log; tail -12 $OUT/pytest_lmhead.log | cut -c1-300
for dual in 1 0; do
  echo "dual=$dual: $(PRL_LMHEAD_DUAL=$dual python scripts/lmhead_fwd_only.py 8 2>&1 | tail -1)"
done | tee $OUT/fwd_ab.txt
echo "== bench (bwd too)"
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 --skip-library > $OUT/lmhead_bench.jsonl 2> $OUT/lmhead_bench.err
grep -E '"tile": "(256x256|default)"' $OUT/lmhead_bench.jsonl | cut -c1-250; grep -E "leading|only" $OUT/lmhead_bench.jsonl | cut -c1-200; tail -3 $OUT/lmhead_bench.err
