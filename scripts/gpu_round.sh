#!/bin/bash
# One GPU-box session: tests, smoke, bench, rocprof.  Everything is logged under gpurun_out/.
# usage: scripts/gpu_round.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env" | tee $OUT/env.log
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))") >> $OUT/env.log 2>&1
echo "== build check" | tee -a $OUT/env.log
python -c "from pipelinerl_amd import _lib; l=_lib.load(); print('libprl abi', l.prl_abi_version())" >> $OUT/env.log 2>&1

echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log

echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log

echo "== bench tiny"
timeout 300 python bench.py --workload tiny --steps 2 --warmup 1 > $OUT/bench_tiny.log 2>&1
echo "bench tiny exit $?" | tee -a $OUT/bench_tiny.log
tail -2 $OUT/bench_tiny.log | cut -c1-600

echo "== kernel sweep"
timeout 600 python scripts/kernel_sweep.py > $OUT/sweep.log 2>&1
echo "sweep exit $?" | tee -a $OUT/sweep.log
tail -40 $OUT/sweep.log

echo "== bench full"
timeout 900 python bench.py --steps 2 --warmup 1 > $OUT/bench_full.log 2>&1
echo "bench full exit $?" | tee -a $OUT/bench_full.log
tail -2 $OUT/bench_full.log | cut -c1-1500

echo "== rocprof stats of the default bench command"
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-weight-sync > $GRAFT_REPO_ROOT/$OUT/rocprof_stats.log 2>&1
echo "rocprof exit $?" | tee -a $GRAFT_REPO_ROOT/$OUT/rocprof_stats.log
cd $GRAFT_REPO_ROOT
find $OUT/prof_stats -name "*kernel_stats*" | head; for f in $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); do head -12 $f; done
# keep the merge-back small: the raw kernel trace can be large
find $OUT/prof_stats -name "*kernel_trace.csv" -size +2M -delete
tail -3 $OUT/rocprof_stats.log | cut -c1-1200
echo "== pmc passes (separate runs, kernel-trace only)"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/scripts/kernel_sweep.py --quick > $GRAFT_REPO_ROOT/$OUT/pmc_$C.log 2>&1; echo "pmc $C exit $?")
done
find $OUT -name "*counter_collection.csv" | head
echo "== done"
