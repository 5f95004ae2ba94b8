#!/bin/bash
# round 3, session b: MX / transpose-read probe, the three fixed tests, first A/B of the triple-plane d hidden
set -u
OUT=gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
scripts/exp/bin/mx_probe > $OUT/mx_probe.txt 2>&1; echo "probe exit $?"
head -12 $OUT/mx_probe.txt; tail -10 $OUT/mx_probe.txt
timeout 600 python -m pytest tests/test_gpu_actor_flow.py tests/test_gpu_bench_contract.py tests/test_gpu_lmhead_fused.py -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/pytest.log
timeout 600 python scripts/lmhead_ab.py --variants 1:4096,0:4096,0:8192,1:8192 --rounds 3 --fwd > $OUT/ab.jsonl 2>&1
echo "ab exit $?"; cat $OUT/ab.jsonl
