#!/bin/bash
set -u
TAG=${1:-ab2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo skip tests
echo "pytest exit $?"; tail -6 $OUT/pytest_lmhead.log
timeout 600 python scripts/lmhead_ab.py --variants 4:8192::keep,0:8192::keep,8:8192::keep,16:8192::keep --rounds 3 > $OUT/lmhead_ab.jsonl 2> $OUT/lmhead_ab.err
echo "ab exit $?"; tail -2 $OUT/lmhead_ab.err
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/lmhead_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (dict, list))})
PY
true
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/fwd_tile_ab.jsonl"):
    d = json.loads(l); print(d["shape"], d["weight"], d["ms"], {k: max(v) for k, v in d["max_abs_diff_vs_default(nlp,ent,lse2)"].items()})
PY
