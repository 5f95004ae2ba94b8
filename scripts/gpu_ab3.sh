#!/bin/bash
set -u
TAG=${1:-ab3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python scripts/exp/lmhead_ps_debug.py 512 3584 512 2048 2>&1 | grep -v amdgpu.ids | head -4
python scripts/exp/lmhead_ps_debug.py 300 192 1088 2048 2>&1 | grep -v amdgpu.ids | head -3
bash scripts/gpu_ab.sh $TAG default,default:2048,default:512 7b,32b
