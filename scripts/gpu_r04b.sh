#!/bin/bash
set -u
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_pipeline.py tests/test_gpu_qwen32b.py tests/test_gpu_actor_flow.py -m gpu -q --maxfail=30 --timeout 600 -p no:cacheprovider -k "bench_json or reference_logprobs or stream or pipeline or actor_flow" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
( time timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-weight-sync --no-e2e --no-transport --no-live-pmc --no-fused-head ) > $OUT/bench_pre.log 2> $OUT/bench_pre.err
echo "bench exit $?"; tail -3 $OUT/bench_pre.err
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
d = json.loads([l for l in open(f"{out}/bench_pre.log") if l.startswith("{")][0])
p = d.get("preprocess_loop") or {}
if "cases" in p:
    for k, c in p["cases"].items():
        print(" pre", k, round(c["us_per_token"], 4), "us/tok", round(c["tokens_per_s"] / 1e6, 2), "Mtok/s planning", round(c["host_planning_frac"], 3), {a: round(b) for a, b in c["host_phase_us_per_chunk"].items()}, {a: round(b) for a, b in c["kernel_us_per_chunk"].items()}, c.get("transfers_per_chunk"))
    print(" pre speedup", p.get("speedup_vs_reference_preprocess_plus_collate"), p.get("speedup_vs_reference_incl_wire"))
else:
    print(" pre", p)
PY
