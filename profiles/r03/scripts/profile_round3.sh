#!/bin/bash
# Runs on the GPU box (through gpurun): everything profiles/<tag>/ of round 3 is distilled from (tools/summarize_profiles.py <tag>
# does the distilling back home).  Counters are collected in their own rocprofv3 runs with --kernel-trace only, as the pool requires.
#   bench*.json      un-profiled bench lines: N = 1 (the driver's command), stream2048, 2 ranks rehearsed on one device
#   stats            rocprofv3 --kernel-trace --stats on bench.py itself
#   req_rd / req_wr  memory-side requests BY SIZE (TCC_EA0_RDREQ / _32B / _64B / _128B, TCC_EA0_WRREQ / _64B): exact HBM-side bytes per
#                    launch instead of FETCH_SIZE's "requests x 64 B"
#   indexed*         the same for the indexed-output pipeline (tools/ab_bench.py indexed=1)
#   stream           kernel stats of bench.py --config stream2048
#   calib            tools/calib/mix2: read / write / copy / mixed-stream ceilings and the chunked-write pattern
#   ab_*.txt         same-box A/Bs of the emit variants (round 2's loop, asynchronous prefetch, vertex-once) and their ablations
#   rank_step*.txt   tools/rank_step.py
# usage: tools/profile_round3.sh <tag>      -> gpurun_out/prof_<tag>/
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/req_rd $OUT/req_wr $OUT/stream $OUT/calib $OUT/indexed $OUT/indexed_rd $OUT/indexed_wr
export TMPDIR=/tmp
cd /tmp
T="timeout -k 10 240"
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
$T python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done"
$T python3 $R/bench.py --config stream2048 > $OUT/bench_stream2048.json 2> $OUT/bench_stream.err
VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo $T python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29719 $R/bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_2rank_one_device_gloo.json 2> $OUT/bench_2rank.err
$T python3 $R/tools/rank_step.py 2 4 8 > $OUT/rank_step.txt 2>&1
$T python3 $R/tools/rank_step.py --comm 8 > $OUT/rank_step_comm.txt 2>&1
echo "rank steps done"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-indexed > $OUT/stats/bench.json 2> $OUT/stats/err.log
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/req_rd -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed > $OUT/req_rd/bench.json 2> $OUT/req_rd/err.log
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/req_wr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed > $OUT/req_wr/bench.json 2> $OUT/req_wr/err.log
echo "soup counters done"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream -- python3 $R/bench.py --config stream2048 --steps 2 --warmup 1 > $OUT/stream/bench.json 2> $OUT/stream/err.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/indexed -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 9 > $OUT/indexed/ab.log 2> $OUT/indexed/err.log
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/indexed_rd -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_rd/err.log
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/indexed_wr -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_wr/err.log
echo "indexed counters done"
$T $R/tools/calib/mix2 4 > $OUT/calib/mix2.jsonl 2> $OUT/calib/err.log
$T $R/tools/calib/mix2 4 chunks > $OUT/calib/chunks.jsonl 2>> $OUT/calib/err.log
$T python3 $R/tools/ab_bench.py "emit_async=0,emit_once=0" "emit_async=1,emit_once=0" "emit_async=1,emit_once=1" "emit_async=0,indexed=1" "emit_async=1,indexed=1" --rounds 9 > $OUT/ab_emit_variants.txt 2>&1
$T python3 $R/tools/ab_bench.py "base" "emit_ablate=1" "emit_ablate=4" "emit_ablate=5" "emit_once=0" "emit_once=0,emit_ablate=1" "emit_once=0,emit_ablate=4" "emit_once=0,emit_ablate=5" --rounds 5 > $OUT/ab_emit_ablation.txt 2>&1
echo "profiles in $OUT"
