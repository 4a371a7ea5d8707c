#!/bin/bash
# Round 6's refresh on the GPU box (through gpurun); tools/summarize_profiles.py r06 distils it into profiles/r06/.
#   pytest_gpu.log            the full GPU suite on the library of this snapshot
#   bench_n1.json             the driver's command: supervisor -> worker; line with roofline, cpu_baseline, box, predicted_scaling, stream2048
#   bench_stream2048.json     BASELINE configs[4] in full, with its CPU leg (stratified sample)
#   bench_2rank_*, bench_world_of_one_comm.json   the N > 1 host path rehearsed on one device (gloo) / through a world-of-one RCCL communicator
#   rank_rehearsal_all*.txt   every rank of N = 2 / 4 / 8, the shipped N > 1 default and the fallback, c % N and the balanced cut
#   stats / stats_s1          rocprofv3 --kernel-trace --stats of the bench (--direct: the program itself behind `--`) and of `--streams 1`
#   req_rd / req_wr           memory-side requests BY SIZE (exact HBM-side bytes per launch), `--streams 1`
#   indexed*, stream          the indexed pipeline, config 5
#   placement_probe.txt       six contexts of one library in one process: the emit kernel's time by output buffer
# usage: tools/profile_round6.sh [tag]      -> gpurun_out/prof_<tag>/
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/stats_s1 $OUT/req_rd $OUT/req_wr $OUT/stream $OUT/indexed $OUT/indexed_rd $OUT/indexed_wr
cd $R
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
export TMPDIR=/tmp
T="timeout -k 10 420"
$T python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
$T python3 $R/bench.py --config stream2048 > $OUT/bench_stream2048.json 2> $OUT/bench_stream.err; echo "stream rc=$?"
VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo $T python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29719 $R/bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_2rank_one_device_gloo.json 2> $OUT/bench_2rank.err; echo "2rank rc=$?"
VTMC_BENCH_FORCE_COMM=1 $T python3 $R/bench.py --no-cpu-baseline --no-indexed --no-rehearsal --no-stream-record --no-terrain-record > $OUT/bench_world_of_one_comm.json 2> $OUT/bench_comm.err; echo "comm rc=$?"
$T python3 $R/tools/rank_rehearsal_all.py --json $OUT/rank_rehearsal_all.json 2> $OUT/rank_rehearsal_all.err | grep -v "$F" > $OUT/rank_rehearsal_all.txt; echo "rehearsal rc=$?"
$T python3 $R/tools/rank_rehearsal_all.py --gather-stream main 2> /dev/null | grep -v "$F" > $OUT/rank_rehearsal_all_gather_stream_main.txt
$T python3 $R/tools/placement_probe.py --contexts 6 --rounds 7 2>&1 | grep -v "$F" > $OUT/placement_probe_refresh.txt
echo "lines and rehearsals done"
echo "profiles in $OUT (part 1: suite, lines, rehearsals); tools/profile_round6_part2.sh for the profiler passes"
