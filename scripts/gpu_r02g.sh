#!/bin/bash
set -u
OUT=gpurun_out/r02g
mkdir -p $OUT
for h in 3584 3648 3520 4096 3712; do
  echo "hidden=$h"; timeout 300 python scripts/lmhead_fused_bench.py --hidden $h --iters 3 --skip-bwd --skip-library 2>/dev/null | grep -E '"tile": "(256x256|256)", "nsplit": "default"' | cut -c1-200
done | tee $OUT/stride_probe.txt
