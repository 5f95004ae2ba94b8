#!/bin/bash
set -u
OUT=gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp
echo "== re-run of the two fixed tests + new pipeline tests"
timeout 900 python -m pytest tests/test_gpu_native_ddp.py tests/test_gpu_pipeline.py -q --maxfail=10 --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "exit $?"; tail -15 $OUT/pytest.log | cut -c1-300
echo "== e2e learner, Qwen2.5-7B shape, bs 16 x 8192, micro-batch 1"
for head in "" "--split-head" "--fused-head"; do
  tag=$(echo "${head:-fp32}" | tr -d '-')
  timeout 600 python scripts/e2e_learner_bench.py --model 7b --batch-size 16 --seq-len 8192 --micro-batch 1 --fused --steps 1 --warmup 1 $head --out $OUT/e2e_7b_$tag.json > $OUT/e2e_7b_$tag.log 2>&1
  echo "head=$tag exit $?"; tail -1 $OUT/e2e_7b_$tag.log | cut -c1-700
done
echo "== e2e learner, Qwen2.5-0.5B shape, bs 512 x 2048"
for head in "--split-head" "--fused-head"; do
  tag=$(echo "${head}" | tr -d '-')
  timeout 600 python scripts/e2e_learner_bench.py --model 0p5b --fused --steps 1 --warmup 1 $head --out $OUT/e2e_0p5b_$tag.json > $OUT/e2e_0p5b_$tag.log 2>&1
  echo "head=$tag exit $?"; tail -1 $OUT/e2e_0p5b_$tag.log | cut -c1-700
done
