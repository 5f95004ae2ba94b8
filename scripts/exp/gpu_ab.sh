#!/bin/bash
# A/B session: scripts/lmhead_fwd_tile_ab.py with the tile list of $2, shapes $3
set -u
TAG=${1:-ab}
TILES=${2:-default,default:512,256x384}
SHAPES=${3:-7b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/lmhead_fwd_tile_ab.py --rounds 3 --iters 4 --shapes $SHAPES --tiles $TILES > $OUT/fwd_tile_ab.jsonl 2> $OUT/fwd_tile_ab.err
echo "ab exit $?"; tail -3 $OUT/fwd_tile_ab.err
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/fwd_tile_ab.jsonl"):
    d = json.loads(l); print(d["shape"], d["weight"], d["ms"], {k: max(v) for k, v in d["max_abs_diff_vs_default(nlp,ent,lse2)"].items()})
PY
