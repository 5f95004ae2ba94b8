#!/bin/bash
set -u
OUT=gpurun_out/r02f
mkdir -p $OUT
for rep in 1 2; do for e in 0 1 2 3 4 8 10; do
  echo "exp=$e: $(PRL_LMHEAD_EXP=$e PRL_LMHEAD_TILE=256x256 python scripts/lmhead_fwd_only.py 8 2>&1 | tail -1)"
done; done | tee $OUT/fwd_exp.txt
