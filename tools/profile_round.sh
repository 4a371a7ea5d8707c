#!/bin/bash
# Runs on the GPU box (through gpurun): everything a round's profiles/<tag>/ is distilled from
# (tools/summarize_profiles.py <tag> does the distilling back home).
#   stats      rocprofv3 --kernel-trace --stats on bench.py itself (the command the driver runs)
#   fetch/write  the two PMC passes the microarch guide prescribes (FETCH_SIZE and WRITE_SIZE cannot share one)
#   indexed*   kernel stats + the two PMC passes of the indexed-output pipeline (tools/ab_bench.py indexed=1)
#   stream     kernel stats of bench.py --config stream2048
#   calib      tools/calib/calib: 40-byte-row FETCH_SIZE calibration (with its own PMC pass) + read/write mix ceiling
#   bench*.json  un-profiled bench lines: N = 1, stream2048, 2 ranks rehearsed on one device; tools/rank_step.py
# usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/stream $OUT/calib $OUT/indexed $OUT/indexed_fetch $OUT/indexed_write
export TMPDIR=/tmp
cd /tmp
T="timeout -k 10 240"
$T python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
$T python3 $R/bench.py --config stream2048 > $OUT/bench_stream2048.json 2> $OUT/bench_stream.err
VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo $T python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29719 $R/bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_2rank_one_device_gloo.json 2> $OUT/bench_2rank.err
$T python3 $R/tools/rank_step.py 2 4 8 > $OUT/rank_step.txt 2>&1
$T python3 $R/tools/rank_step.py --comm 8 > $OUT/rank_step_comm.txt 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-indexed > $OUT/stats/bench.json 2> $OUT/stats/err.log
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed > $OUT/fetch/bench.json 2> $OUT/fetch/err.log
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed > $OUT/write/bench.json 2> $OUT/write/err.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream -- python3 $R/bench.py --config stream2048 --steps 2 --warmup 1 > $OUT/stream/bench.json 2> $OUT/stream/err.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/indexed -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 9 > $OUT/indexed/ab.log 2> $OUT/indexed/err.log
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/indexed_fetch -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_fetch/err.log
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/indexed_write -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_write/err.log
$T $R/tools/calib/calib mix > $OUT/calib/mix.json 2> $OUT/calib/err.log
$T $R/tools/calib/calib rows > $OUT/calib/rows.json 2>> $OUT/calib/err.log
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib/pmc -- $R/tools/calib/calib rows > $OUT/calib/rows_under_pmc.json 2>> $OUT/calib/err.log
echo "profiles in $OUT"
