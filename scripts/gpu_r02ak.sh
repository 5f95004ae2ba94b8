#!/bin/bash
# full validation: every GPU test, smoke, the default bench line, rocprofv3 kernel stats of the same command
set -u
OUT=gpurun_out/r02ak
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?"; tail -3 $OUT/smoke.log | cut -c1-300
echo "== bench default (N=1)"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "exit $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02ak/bench_default.json").read().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps","warmup")})
print("roofline", d["roofline"])
print("roofline_mfma", {k:v for k,v in (d["roofline_mfma"] or {}).items() if k not in ("note","config")})
print("kernels", {k:(round(v["avg_us"],1), round(v.get("hbm_frac",0),3)) for k,v in d["kernels"].items()})
print("wsync", {k:d["weight_sync"].get(k) for k in ("median_ms","gbytes","error")})
PY
tail -3 $OUT/bench_default.err
echo "== rocprofv3 kernel stats of the default bench command"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o stats -- python $R/bench.py --no-cpu-baseline --no-weight-sync > $R/$OUT/rocprof_bench.log 2>&1; echo "rocprof exit $?")
f=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -12 "$f" | cut -c1-260
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.db" -delete
