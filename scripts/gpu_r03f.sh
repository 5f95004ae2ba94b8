#!/bin/bash
set -u
OUT=gpurun_out/r03f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py tests/test_gpu_fused_head_ddp.py -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/pytest.log
timeout 600 python scripts/lmhead_ab.py --variants 3:8192,2:8192,0:8192,0:4096 --rounds 3 > $OUT/ab.jsonl 2>&1
echo "ab exit $?"; cat $OUT/ab.jsonl
