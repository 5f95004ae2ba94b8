#!/bin/bash
set -u
OUT=gpurun_out/r02al
mkdir -p $OUT
export TMPDIR=/tmp
echo "== e2e learner, Qwen2.5-0.5B shape, bs 512 x 2048"
for head in "--split-head" "--fused-head"; do
  tag=$(echo "${head}" | tr -d '-')
  timeout 600 python scripts/e2e_learner_bench.py --model 0p5b --fused --steps 1 --warmup 1 $head --out $OUT/e2e_0p5b_$tag.json > $OUT/e2e_0p5b_$tag.log 2>&1
  echo "head=$tag exit $?"; tail -1 $OUT/e2e_0p5b_$tag.log | cut -c1-600
done
