#!/bin/bash
# One GPU-box session: the whole -m gpu suite, smoke, the default bench line (live 7B model-in-the-loop step included),
# rocprofv3 kernel stats of the same command.  Everything is logged under gpurun_out/<tag>/.
# usage: /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash scripts/gpu_round.sh r03z'
set -u
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) > $OUT/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
( time timeout 1500 python bench.py --steps 5 --warmup 2 ) > $OUT/bench_full.log 2> $OUT/bench_full.err
echo "bench exit $?" | tee -a $OUT/bench_full.log
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
d = json.loads([l for l in open(out + "/bench_full.log") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"], 4))
m = d["roofline_mfma"]
print("head fwd ms", round(m["ms_per_launch"], 2), "keeping fwd ms", m.get("forward_keeping_logits", {}).get("ms"), "bwd ms", round(m["backward"]["ms"], 2),
      "recomputing bwd ms", m.get("backward_recompute", {}).get("ms"), "mixed fwd", m.get("forward_mixed_precision", {}).get("ms"))
e = d["e2e"]
print({k: e.get(k) for k in ("s_per_step", "samples_per_s", "peak_memory_GB", "source", "error")})
print({k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-weight-sync --no-e2e --no-transport --no-live-pmc > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
# the summary for profiles/: names shortened, ALL numeric columns kept (a width cut lost the dominant kernel's numbers in round 3)
cd $GRAFT_REPO_ROOT; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do python scripts/kernel_stats_summary.py $f $OUT/bench_kernel_stats.csv; head -8 $OUT/bench_kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -delete
