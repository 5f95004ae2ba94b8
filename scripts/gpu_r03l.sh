#!/bin/bash
# N > 1 control flow of bench.py, dry-run: two ranks on ONE GPU over gloo (no RCCL between devices exists on a 1-GPU box)
set -u
OUT=gpurun_out/r03l
mkdir -p $OUT
export TMPDIR=/tmp
PRL_BENCH_SHARE_DEVICE=1 PRL_BENCH_WORKLOAD=0p5b_grpo_bs512_seq2048 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo > $OUT/bench_2ranks_gloo.log 2> $OUT/bench_2ranks_gloo.err
echo "exit $?"; grep "^{" $OUT/bench_2ranks_gloo.log | cut -c1-900; tail -5 $OUT/bench_2ranks_gloo.err
