#!/bin/bash
set -u
OUT=gpurun_out/r03d
mkdir -p $OUT
export TMPDIR=/tmp
PRL_LMHEAD_PRECISION=f16_fp8 timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q --timeout 600 -p no:cacheprovider > $OUT/pytest_mx.log 2>&1
echo "pytest (mx) exit $?"; tail -15 $OUT/pytest_mx.log
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_bf16.log 2>&1
echo "pytest (bf16x2) exit $?"; tail -4 $OUT/pytest_bf16.log
PRL_LMHEAD_PRECISION=f16_fp8 timeout 600 python scripts/lmhead_ab.py --variants 0:4096,0:8192 --rounds 3 --fwd > $OUT/ab_mx.jsonl 2>&1
echo "ab exit $?"; cat $OUT/ab_mx.jsonl
