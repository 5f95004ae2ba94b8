#!/bin/bash
set -u
OUT=gpurun_out/r03c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/lmhead_mx_check.py > $OUT/mx_check.jsonl 2>&1; echo "mx exit $?"; cat $OUT/mx_check.jsonl | tail -20
timeout 600 python -m pytest tests/test_gpu_actor_flow.py tests/test_gpu_bench_contract.py -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/pytest.log
