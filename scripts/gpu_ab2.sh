#!/bin/bash
set -u
TAG=${1:-ab2}
VARIANTS=${2:-16:8192::keep,0:8192::keep,4:8192::keep,0:8192,16:8192}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py tests/test_gpu_qwen32b.py tests/test_gpu_fused_head_ddp.py -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider -k "not bucket and not tp2" > $OUT/pytest_lmhead.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/pytest_lmhead.log
timeout 600 python scripts/lmhead_ab.py --variants $VARIANTS --rounds 3 --fwd > $OUT/lmhead_ab.jsonl 2> $OUT/lmhead_ab.err
echo "ab exit $?"; tail -2 $OUT/lmhead_ab.err
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/lmhead_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (dict, list))})
PY
