#!/bin/bash
set -u
OUT=gpurun_out/r02i
mkdir -p $OUT
export TMPDIR=/tmp
echo "== full gpu suite"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log | cut -c1-400
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "exit $?"; tail -2 $OUT/smoke.log
