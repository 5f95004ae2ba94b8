#!/bin/bash
set -u
OUT=gpurun_out/r03m
mkdir -p $OUT
export TMPDIR=/tmp
# bits: 0 = shipped; 4 = recompute without global stores; 8 = recompute without the plane epilogue (timing ablations: results wrong)
timeout 600 python scripts/lmhead_ab.py --variants 0:8192,4:8192,8:8192 --rounds 3 2>&1 | grep bits | python -c "import sys,json; [print(d['bits'], {k:round(d[k],2) for k in ('ms_min','ms_dh_only_min','ms_dw_only_min')}) for d in map(json.loads, sys.stdin)]" | tee $OUT/ablate.txt
