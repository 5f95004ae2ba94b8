#!/bin/bash
set -u
OUT=gpurun_out/r03n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/pytest.log
timeout 600 python scripts/lmhead_ab.py --variants 0:8192,0:4096,0:8192:f16_fp8 --rounds 3 2>&1 | grep bits | python -c "import sys,json; [print(d['bits'], d['chunk_rows'], d['precision'], {k:(round(d[k],2) if 'ms' in k else d[k]) for k in ('ms_min','ms_dh_only_min','ms_dw_only_min','d_hidden_vs_round2','d_weight_vs_round2')}) for d in map(json.loads, sys.stdin)]" | tee $OUT/ab.txt
