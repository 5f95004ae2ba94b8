#!/bin/bash
# The N > 1 control flow of bench.py on a 1-GPU box: N ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one GPU).
# Covers sharding by whole groups, the per-rank step, the statistics all-gather (the step must cover all 4096 sequences) and the
# weight-sync probe's error handling; not a measurement.   usage: gpurun -- 'bash scripts/bench_dry_run_2ranks.sh <tag> [workload] [ranks]'
set -u
TAG=${1:-dry}
WL=${2:-7b_grpo_bs4096_seq8192}
N=${3:-2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PRL_BENCH_SHARE_DEVICE=1 PRL_BENCH_FORCE_WSYNC=1 PRL_BENCH_WSYNC_TIMEOUT=60
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 1 --warmup 0 \
  --backend gloo --workload $WL --detail-out $OUT/dry_run_${N}ranks_detail.json > $OUT/dry_run_${N}ranks.log 2> $OUT/dry_run_${N}ranks.err
echo "exit $?"; grep '^{' $OUT/dry_run_${N}ranks.log | tail -1 | cut -c1-2600; tail -3 $OUT/dry_run_${N}ranks.err
