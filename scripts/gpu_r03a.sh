#!/bin/bash
# round 3, first GPU session: the whole -m gpu suite on the new tree + smoke + the default bench line (live e2e)
set -u
OUT=gpurun_out/r03a
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) > $OUT/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
( time timeout 1500 python bench.py --steps 5 --warmup 2 ) > $OUT/bench_full.log 2> $OUT/bench_full.err
echo "bench exit $?" | tee -a $OUT/bench_full.log
tail -c 6000 $OUT/bench_full.log
tail -5 $OUT/bench_full.err
