#!/bin/bash
# The N > 1 control flow of bench.py on a 1-GPU box: two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one GPU).
# Covers sharding, the stats all-gather, the per-rank ref_logprob probe and the weight-sync probe's error handling; not a measurement.
# usage: gpurun -- 'bash scripts/bench_dry_run_2ranks.sh <tag> [workload]'
set -u
TAG=${1:-dry}
WL=${2:-7b_grpo_bs4096_seq8192}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PRL_BENCH_SHARE_DEVICE=1 PRL_BENCH_FORCE_WSYNC=1 PRL_BENCH_WSYNC_TIMEOUT=60
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 0 \
  --backend gloo --workload $WL --detail-out $OUT/dry_run_detail.json > $OUT/dry_run.log 2> $OUT/dry_run.err
echo "exit $?"; grep '^{' $OUT/dry_run.log | head -1 | cut -c1-1500; tail -3 $OUT/dry_run.err
