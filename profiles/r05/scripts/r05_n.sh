#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_n
mkdir -p $OUT
cd $R
PYTHONFAULTHANDLER=1 VTMC_BENCH_DEBUG=1 VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29653 bench.py --gpus 2 --grid 256 --steps 3 --warmup 1 > $OUT/two_rank.out 2> $OUT/two_rank.err
echo "rc=$?"; grep -n "Fatal\|bench.py\[" $OUT/two_rank.err | head -40
PYTHONFAULTHANDLER=1 VTMC_BENCH_DEBUG=1 VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29654 bench.py --gpus 2 --grid 256 --steps 3 --warmup 1 --streams 1 > $OUT/two_rank_s1.out 2> $OUT/two_rank_s1.err
echo "streams 1 rc=$?"; grep -n "Fatal\|File \"/root/repo" $OUT/two_rank_s1.err | head -20
