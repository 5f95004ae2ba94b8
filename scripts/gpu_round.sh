#!/bin/bash
# One GPU-box session: the whole -m gpu suite, smoke, the DRIVER's bench command (default legs only), rocprofv3 kernel stats of the same
# timed region, then `bench.py --detail` (side measurements -> detail file) and the other two BASELINE workloads.
# Everything is logged under gpurun_out/<tag>/.   usage: gpurun --timeout 3000 -- 'bash scripts/gpu_round.sh r06a [skip-tests]'
set -u
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) > $OUT/env.log 2>&1
if [ "${2:-}" != "skip-tests" ]; then
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider --durations=8 > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
  tail -6 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
# exactly what the driver runs at round end
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $OUT/bench_default_detail.json ) > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "bench exit $?" | tee -a $OUT/bench_default.err
tail -4 $OUT/bench_default.err
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
text = [l for l in open(out + "/bench_default.log") if l.startswith("{")][-1]
d = json.loads(text)
print("line bytes", len(text), {k: d[k] for k in ("value", "ms_per_step", "value_skip_unlabelled", "wall_s")}, "roofline", round(d["roofline"]["frac"], 4), d["roofline"]["avg_us"])
print(" cpu_baseline", d["cpu_baseline"], "\n weight_sync", d["weight_sync"], "\n hbm_frac", d["hbm_frac"], "skipped", d.get("skipped"))
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weight-sync --no-live-pmc --skip-unlabelled-steps 0 --detail-out $OUT/bench_rocprof_detail.json > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
# the summary for profiles/: names shortened, ALL numeric columns kept (a width cut lost the dominant kernel's numbers in round 3)
cd $GRAFT_REPO_ROOT; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do python scripts/kernel_stats_summary.py $f $OUT/bench_kernel_stats.csv; head -8 $OUT/bench_kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -delete
grep '^{' $OUT/rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
if [ "${3:-}" == "no-detail" ]; then exit 0; fi
# side measurements (never on the line): MFMA head, model-in-the-loop step, reference-policy head, preprocessor loop, configs[1] pipeline, transport
( time timeout 1500 python bench.py --steps 3 --warmup 1 --detail --detail-out $OUT/bench_detail.json ) > $OUT/bench_detail.log 2> $OUT/bench_detail.err
echo "bench --detail exit $?"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
try:
    d = json.load(open(out + "/bench_detail.json"))
except Exception as e:
    print("no detail file", e); sys.exit(0)
m = d.get("roofline_mfma") or {}
print("head fwd ms", m.get("ms_per_launch"), "keeping fwd ms", m.get("forward_keeping_logits", {}).get("ms"), "bwd ms", (m.get("backward") or {}).get("ms"),
      "recomputing bwd ms", m.get("backward_recompute", {}).get("ms"))
e = d.get("e2e") or {}
print({k: e.get(k) for k in ("s_per_step", "samples_per_s", "peak_memory_GB", "source", "error")})
print({k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
p = d.get("preprocess_loop") or {}
for k, c in (p.get("cases") or {}).items():
    print(" pre", k, round(c["us_per_token"], 4), "us/tok planning", round(c["host_planning_frac"], 3))
pl = d.get("pipeline") or {}
print(" pipeline", {k: pl.get(k) for k in ("samples_per_s", "s_per_step", "busy_frac", "error")})
PY
# the other two BASELINE workloads (configs[4]: 32B, KL on; configs[1]: 0.5B)
( time timeout 600 python bench.py --workload 32b_grpo_kl_bs4096_seq8192 --steps 2 --warmup 1 --detail-out $OUT/bench_32b_detail.json ) > $OUT/bench_32b.log 2> $OUT/bench_32b.err
echo "bench 32b exit $?"
( time timeout 300 python bench.py --workload 0p5b_grpo_bs512_seq2048 --steps 5 --warmup 2 --detail-out $OUT/bench_0p5b_detail.json ) > $OUT/bench_0p5b.log 2> $OUT/bench_0p5b.err
echo "bench 0p5b exit $?"
grep -h '^{' $OUT/bench_32b.log $OUT/bench_0p5b.log | cut -c1-600
