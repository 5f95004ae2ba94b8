#!/bin/bash
set -u
OUT=gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/pack_bench.py > $OUT/pack.txt 2>&1; cat $OUT/pack.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_pipeline.py -q -x --timeout 600 -p no:cacheprovider -k "collate or pack or ragged or pipeline or replay" > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/pytest.log
