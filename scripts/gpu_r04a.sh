#!/bin/bash
# Round-4 first GPU session: the -m gpu suite (new 32B / ref-logprob / full-size tests included), then the KL-on 32B workload and
# the 0.5B workload of bench.py.  usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r04a.sh r04a'
set -u
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; df -h /dev/shm | tail -1) > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 --timeout 600 -p no:cacheprovider --durations=12 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
( time timeout 600 python bench.py --workload 32b_grpo_kl_bs4096_seq8192 --steps 2 --warmup 1 --no-cpu-baseline ) > $OUT/bench_32b.log 2> $OUT/bench_32b.err
echo "bench 32b exit $?"; tail -3 $OUT/bench_32b.err
( time timeout 300 python bench.py --workload 0p5b_grpo_bs512_seq2048 --steps 5 --warmup 2 ) > $OUT/bench_0p5b.log 2> $OUT/bench_0p5b.err
echo "bench 0p5b exit $?"; tail -3 $OUT/bench_0p5b.err
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for name in ("bench_32b", "bench_0p5b"):
    try:
        d = json.loads([l for l in open(f"{out}/{name}.log") if l.startswith("{")][0])
    except Exception as e:
        print(name, "no line", e); continue
    print(name, {k: d[k] for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"], 4), "avg_us", round(d["roofline"]["avg_us"], 1))
    m = d.get("roofline_mfma") or {}
    print(" head", {k: (round(m[k], 2) if isinstance(m.get(k), float) else m.get(k)) for k in ("ms_per_launch", "achieved", "error")}, "bwd", (m.get("backward") or {}).get("ms"), "rec", (m.get("backward_recompute") or {}).get("ms"))
    print(" ref_logprob", json.dumps(d.get("ref_logprob"))[:900])
    p = d.get("preprocess_loop") or {}
    if "cases" in p:
        for k, c in p["cases"].items():
            print(" pre", k, round(c["us_per_token"], 4), "us/tok planning", round(c["host_planning_frac"], 3), {a: round(b) for a, b in c["host_phase_us_per_chunk"].items()}, {a: round(b) for a, b in c["kernel_us_per_chunk"].items()})
        print(" pre speedup", p.get("speedup_vs_reference_preprocess_plus_collate"), p.get("speedup_vs_reference_incl_wire"))
    else:
        print(" pre", p)
    print(" wsync", d.get("weight_sync"))
    print(" kernels", {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
