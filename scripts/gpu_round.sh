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
      "recomputing bwd ms", m.get("backward_recompute", {}).get("ms"))
e = d["e2e"]
print({k: e.get(k) for k in ("s_per_step", "samples_per_s", "peak_memory_GB", "source", "error")})
print({k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
p = d.get("preprocess_loop") or {}
for k, c in (p.get("cases") or {}).items():
    print(" pre", k, round(c["us_per_token"], 4), "us/tok planning", round(c["host_planning_frac"], 3), "consumer", {a: round(b, 4) for a, b in (c.get("consumer") or {}).items() if isinstance(b, float)})
print(" pre speedup", p.get("speedup_vs_reference_preprocess_plus_collate"), "ref_logprob", {k: (round(v["old_ms"], 2), round(v["fused_ms"], 2)) for k, v in ((d.get("ref_logprob") or {}).get("heads") or {}).items()})
print(" cpu_baseline", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], "port", (d["cpu_baseline"].get("port") or {}).get("value"), "cores", d["cpu_baseline"]["cores"])
pl = d.get("pipeline") or {}
print(" pipeline", {k: pl.get(k) for k in ("samples_per_s", "s_per_step", "busy_frac", "error")}, "budget8192", (pl.get("pack_budget_8192") or {}).get("samples_per_s"),
      "wsync under load", (pl.get("weight_sync_under_load_ms") or {}).get("median"))
for k, h in ((d.get("ref_logprob") or {}).get("heads") or {}).items():
    print(" ref head", k, "err vs fp64", h.get("max_abs_error_vs_fp64"), "equal-accuracy library", (h.get("old_equal_accuracy") or {}))
print(" value_e2e", d.get("value_e2e"), "wsync transport", (d.get("weight_sync") or {}).get("transport"))
PY
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-weight-sync --no-e2e --no-transport --no-live-pmc --no-pipeline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
# the summary for profiles/: names shortened, ALL numeric columns kept (a width cut lost the dominant kernel's numbers in round 3)
cd $GRAFT_REPO_ROOT; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do python scripts/kernel_stats_summary.py $f $OUT/bench_kernel_stats.csv; head -8 $OUT/bench_kernel_stats.csv; done
find $OUT/prof -name "*kernel_trace.csv" -delete
# the other two BASELINE workloads (configs[4]: 32B, KL on; configs[1]: 0.5B)
( time timeout 600 python bench.py --workload 32b_grpo_kl_bs4096_seq8192 --steps 2 --warmup 1 --no-preprocess-loop --no-transport --no-pipeline ) > $OUT/bench_32b.log 2> $OUT/bench_32b.err
echo "bench 32b exit $?"
( time timeout 300 python bench.py --workload 0p5b_grpo_bs512_seq2048 --steps 5 --warmup 2 --no-preprocess-loop --no-transport --no-pipeline ) > $OUT/bench_0p5b.log 2> $OUT/bench_0p5b.err
echo "bench 0p5b exit $?"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for name in ("bench_32b", "bench_0p5b"):
    try:
        d = json.loads([l for l in open(f"{out}/{name}.log") if l.startswith("{")][0])
    except Exception as e:
        print(name, "no line", e); continue
    m = d.get("roofline_mfma") or {}
    print(name, round(d["value"], 1), "samples/s roofline", round(d["roofline"]["frac"], 4), "head fwd", m.get("ms_per_launch"), "bwd", (m.get("backward") or {}).get("ms"),
          "ref_logprob", {k: (round(v["old_ms"], 2), round(v["fused_ms"], 2)) for k, v in ((d.get("ref_logprob") or {}).get("heads") or {}).items()},
          "wsync", (d.get("weight_sync") or {}).get("median_ms"), "cpu", (d.get("cpu_baseline") or {}).get("kind"))
PY
