#!/bin/bash
# Round 6, first GPU call: the GPU suite, the driver's command, the rank rehearsal of every rank, config 5 -- and LAST, chained by && (a step that
# times out ends the call), the exit-hang probes of VERDICT r05 item 3.
TAG=${1:-r06a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
T="timeout -k 10 420"
$T python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
$T python3 $R/tools/rank_rehearsal_all.py --json $OUT/rank_rehearsal_all.json 2> $OUT/rank_rehearsal_all.err | grep -v "$F" > $OUT/rank_rehearsal_all.txt; echo "rehearsal rc=$?"
$T python3 $R/bench.py --config stream2048 > $OUT/bench_stream2048.json 2> $OUT/bench_stream.err; echo "stream rc=$?"
VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo $T python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29719 $R/bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_2rank_one_device_gloo.json 2> $OUT/bench_2rank.err; echo "2rank rc=$?"
VTMC_BENCH_FORCE_COMM=1 $T python3 $R/bench.py --no-cpu-baseline --no-indexed --no-rehearsal --no-stream-record > $OUT/bench_world_of_one_comm.json 2> $OUT/bench_comm.err; echo "comm rc=$?"
# exit-hang probes: each under its own timeout, the chain ends at the first that does not come back
P=$R/tools/calib/cumask_exit_probe
H=$R/host/_build/host_selftest
mkdir -p $OUT/selftest
{
  X="timeout -k 5 90"
  $X $P plain && echo "probe plain rc=0" &&
  $X $P keep && echo "probe keep rc=0" &&
  $X $P r6 && echo "probe r6 rc=0" &&
  VTMC_TEST_MAIN_STREAM_OWN_QUEUE=1 $X $H --gpu $OUT/selftest && echo "host_selftest --gpu, main stream on its own queue, streams parked: rc=0" &&
  VTMC_TEST_MAIN_STREAM_OWN_QUEUE=1 VTMC_STREAM_POOL=0 $X $H --gpu $OUT/selftest && echo "host_selftest --gpu, main stream on its own queue, streams destroyed in round 6's order: rc=0" &&
  $X $P r5 && echo "probe r5 rc=0"
  echo "exit-hang chain ended with rc=$?"
} > $OUT/exit_hang_probes.txt 2>&1
cat $OUT/exit_hang_probes.txt | tail -12
echo "profiles in $OUT"
