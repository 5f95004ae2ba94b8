#!/bin/bash
set -u
TAG=${1:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -m gpu -q --maxfail=40 --timeout 600 -p no:cacheprovider -k "small_ragged or split_count or qwen7b" > $OUT/pytest_lmhead.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_lmhead.log
tail -15 $OUT/pytest_lmhead.log
timeout 600 python scripts/lmhead_fwd_tile_ab.py --rounds 3 --iters 4 > $OUT/fwd_tile_ab.jsonl 2> $OUT/fwd_tile_ab.err
echo "ab exit $?"; cat $OUT/fwd_tile_ab.jsonl; tail -3 $OUT/fwd_tile_ab.err
timeout 600 python scripts/kernel_sweep.py > $OUT/kernel_sweep.txt 2>&1
echo "sweep exit $?"; grep -i "fused K1\|copy" $OUT/kernel_sweep.txt
