#!/bin/bash
set -u
OUT=gpurun_out/r02z
mkdir -p $OUT
export TMPDIR=/tmp
echo "== e2e learner, Qwen2.5-7B shape, bs 16 x 8192, micro-batch 1"
for head in "--split-head" "--fused-head"; do
  tag=$(echo "${head}" | tr -d '-')
  timeout 600 python scripts/e2e_learner_bench.py --model 7b --batch-size 16 --seq-len 8192 --micro-batch 1 --fused --steps 1 --warmup 1 $head --out $OUT/e2e_7b_$tag.json > $OUT/e2e_7b_$tag.log 2>&1
  echo "head=$tag exit $?"; tail -1 $OUT/e2e_7b_$tag.log | cut -c1-700
done
