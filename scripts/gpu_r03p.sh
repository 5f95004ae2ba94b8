#!/bin/bash
set -u
OUT=gpurun_out/r03p
mkdir -p $OUT
export TMPDIR=/tmp
for SEG in 64 16 32 128 256 100000 64; do
  echo -n "seg $SEG: "
  PRL_LMHEAD_SEG=$SEG timeout 300 python scripts/lmhead_ab.py --variants 0:8192 --rounds 3 2>&1 | grep bits | python -c "import sys,json; [print({k:round(d[k],2) for k in ('ms_min','ms_dh_only_min','ms_dw_only_min')}) for d in map(json.loads, sys.stdin)]"
done | tee $OUT/seg.txt
