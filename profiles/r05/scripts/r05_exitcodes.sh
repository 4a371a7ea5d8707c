#!/bin/bash
# exit codes of every way bench.py is started (a crash in tear-down after the JSON line would still fail the driver's run)
OUT=gpurun_out/r05_rc; mkdir -p $OUT
run() { name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? lines=$(wc -l < $OUT/$name.json)"; }
run n1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 3
run n1_min timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline
run n1_s1 timeout -k 10 300 python bench.py --streams 1 --steps 5 --warmup 1 --no-cpu-baseline
VTMC_BENCH_FORCE_COMM=1 run comm1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-indexed
VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo run gloo2 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 2 --steps 20 --warmup 3
VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo run gloo4 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29732 bench.py --gpus 4 --steps 10 --warmup 2
run torchrun1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline
run stream timeout -k 10 300 python bench.py --config stream2048 --no-cpu-baseline
grep -l "Segmentation\|core dumped\|Aborted\|Traceback" $OUT/*.err || echo "no crash text in any stderr"
