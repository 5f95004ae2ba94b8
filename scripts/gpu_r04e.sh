#!/bin/bash
set -u
TAG=${1:-r04e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/lmhead_fwd_tile_ab.py --rounds 3 --iters 4 --shapes 7b --tiles default,256x384:2,256x384:3,256x384:4 > $OUT/fwd_tile_ab.jsonl 2> $OUT/fwd_tile_ab.err
echo "ab exit $?"; cat $OUT/fwd_tile_ab.jsonl; tail -3 $OUT/fwd_tile_ab.err
