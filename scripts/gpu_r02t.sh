#!/bin/bash
set -u
OUT=gpurun_out/r02t
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_planes.py tests/test_gpu_fused_head_ddp.py tests/test_gpu_lmhead_fused.py tests/test_gpu_fullvocab.py tests/test_gpu_lm_head.py tests/test_gpu_parity.py -q --maxfail=30 --timeout 400 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "exit $?"; tail -12 $OUT/pytest.log | cut -c1-300
timeout 300 python scripts/planes_bench.py 2>&1 | tee $OUT/planes_bench.jsonl | cut -c1-250
