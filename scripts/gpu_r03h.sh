#!/bin/bash
set -u
OUT=gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
for GM in 1 2 3 4 8; do
  echo "== dW raster group $GM"
  PRL_LMHEAD_DW_GROUP=$GM timeout 300 python scripts/lmhead_ab.py --variants 0:8192 --rounds 3 2>&1 | tee -a $OUT/raw.txt | grep bits | python -c "import sys,json; [print({k:d[k] for k in ('ms_min','ms_dh_only_min','ms_dw_only_min','d_weight_vs_round2')}) for d in map(json.loads, sys.stdin)]"
done > $OUT/raster.txt 2>&1
cat $OUT/raster.txt
