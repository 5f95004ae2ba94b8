#!/bin/bash
# Round 2, GPU session A: new full-vocabulary parity tests, whole gpu suite, smoke, first run of the fused head.
set -u
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0), torch.cuda.device_count())") > $OUT/env.log 2>&1
python -c "from pipelinerl_amd import _lib; l=_lib.load(); print('libprl abi', l.prl_abi_version())" >> $OUT/env.log 2>&1
cat $OUT/env.log | tail -3

echo "== pytest full-vocab parity"
timeout 900 python -m pytest tests/test_gpu_fullvocab.py -q -s --maxfail=60 --timeout 300 -p no:cacheprovider > $OUT/pytest_fullvocab.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_fullvocab.log; grep -E "passed|failed" $OUT/pytest_fullvocab.log | tail -2

echo "== pytest rest of the gpu suite"
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 --timeout 300 -p no:cacheprovider --deselect tests/test_gpu_fullvocab.py --deselect tests/test_gpu_lmhead_fused.py > $OUT/pytest_gpu.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log

echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log

echo "== fused head tests"
timeout 900 python -m pytest tests/test_gpu_lmhead_fused.py -q --maxfail=30 --timeout 300 -p no:cacheprovider > $OUT/pytest_lmhead.log 2>&1
echo "exit $?" | tee -a $OUT/pytest_lmhead.log; tail -25 $OUT/pytest_lmhead.log | cut -c1-300

echo "== fused head bench"
timeout 600 python scripts/lmhead_fused_bench.py --iters 3 > $OUT/lmhead_bench.jsonl 2> $OUT/lmhead_bench.err
echo "exit $?"; cat $OUT/lmhead_bench.jsonl | cut -c1-400; tail -5 $OUT/lmhead_bench.err

echo "== rocprof kernel trace of one default-dispatch parity test (proof of the kernel that ran)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_fullvocab -o t -- python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_fullvocab.py -q -p no:cacheprovider -k "test_default_fused_dispatch_vs_oracle and 152064-grpo_clip" > $GRAFT_REPO_ROOT/$OUT/rocprof_fullvocab.log 2>&1; echo "rocprof exit $?")
find $OUT/prof_fullvocab -name "*kernel_stats.csv" | head -1 | xargs -r head -8
find $OUT -name "*kernel_trace.csv" -size +2M -delete
echo "== done"
