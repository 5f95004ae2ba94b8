#!/bin/bash
set -u
OUT=gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== ddp + bench contract tests"
timeout 900 python -m pytest tests/test_gpu_native_ddp.py tests/test_gpu_bench_contract.py tests/test_gpu_multi.py -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "exit $?"; tail -6 $OUT/pytest.log | cut -c1-300
echo "== bench default (N=1)"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "exit $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02k/bench_default.json").read().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps","warmup")})
print("roofline", d["roofline"])
print("roofline_mfma", {k:v for k,v in (d["roofline_mfma"] or {}).items() if k!="note"})
print("e2e", {k:(d["e2e"] or {}).get(k) for k in ("samples_per_s","s_per_step","loss_forward_fraction","source")})
print("kernels", {k:(round(v["avg_us"],1), round(v.get("hbm_frac",0),3)) for k,v in d["kernels"].items()})
print("cpu", d["cpu_baseline"]["value"], (d["cpu_baseline"].get("reference_autograd") or {}).get("samples_per_s_extrapolated"))
print("wsync", {k:d["weight_sync"].get(k) for k in ("median_ms","gbytes","error")})
PY
tail -3 $OUT/bench_default.err
echo "== bench dry run, 2 ranks on one GPU over gloo (exercises the N>1 control flow)"
PRL_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --workload tiny --backend gloo > $OUT/bench_dry2.json 2> $OUT/bench_dry2.err
echo "exit $?"; tail -1 $OUT/bench_dry2.json | cut -c1-400; tail -2 $OUT/bench_dry2.err | cut -c1-300
echo "== rocprofv3 kernel stats of the default bench command"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o stats -- python $R/bench.py --no-cpu-baseline --no-weight-sync > $R/$OUT/rocprof_bench.log 2>&1; echo "rocprof exit $?")
f=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-260
find $OUT -name "*kernel_trace.csv" -size +1M -delete
