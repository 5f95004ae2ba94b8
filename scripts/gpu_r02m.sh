#!/bin/bash
set -u
OUT=gpurun_out/r02m
mkdir -p $OUT
for rep in 1 2; do for e in 0 128; do
  echo "exp=$e: $(PRL_LMHEAD_EXP=$e python scripts/lmhead_fwd_only.py 8 2>&1 | tail -1)"
done; done | tee $OUT/fwd_stagger.txt
PRL_LMHEAD_EXP=128 timeout 600 python -m pytest tests/test_gpu_lmhead_fused.py -q -k "qwen7b or forward_values" -p no:cacheprovider 2>&1 | tail -3
