#!/bin/bash
# One GPU-box session for the process pipeline: the parity test (tiny shape, four processes) and BASELINE configs[1] as a pipeline.
# usage: /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/gpu_pipeline.sh r05a [steps]'
set -u
TAG=${1:-pipe}
STEPS=${2:-4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; df -h /dev/shm | tail -1) > $OUT/env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_pipeline_procs.py -m gpu -q -x --timeout 800 -p no:cacheprovider > $OUT/pytest_pipeline.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_pipeline.log
tail -40 $OUT/pytest_pipeline.log
( time timeout 900 python scripts/pipeline_cfg1.py --steps $STEPS --out $OUT/pipeline_cfg1.json ) > $OUT/pipeline_cfg1.log 2> $OUT/pipeline_cfg1.err
echo "pipeline exit $?"
tail -5 $OUT/pipeline_cfg1.err
python - "$OUT" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/pipeline_cfg1.json").read())
print(json.dumps(d.get("summary") or d.get("error"), indent=1)[:6000])
PY
