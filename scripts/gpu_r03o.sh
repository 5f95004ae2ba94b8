#!/bin/bash
set -u
OUT=gpurun_out/r03o
mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
for which in prev cur; do
  if [ $which = prev ]; then export PRL_LIB=$PWD/pipelinerl_amd/lib/libprl_prev.so; else unset PRL_LIB; fi
  echo "== $which (round $round)"
  timeout 600 python scripts/lmhead_ab.py --variants 0:8192 --rounds 3 2>&1 | grep bits | python -c "import sys,json; [print({k:round(d[k],2) for k in ('ms_min','ms_dh_only_min','ms_dw_only_min')}) for d in map(json.loads, sys.stdin)]"
done; done | tee $OUT/prev_vs_cur.txt
unset PRL_LIB
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/pytest.log
