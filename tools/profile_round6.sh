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
VTMC_BENCH_FORCE_COMM=1 $T python3 $R/bench.py --no-cpu-baseline --no-indexed --no-rehearsal --no-stream-record > $OUT/bench_world_of_one_comm.json 2> $OUT/bench_comm.err; echo "comm rc=$?"
$T python3 $R/tools/rank_rehearsal_all.py --json $OUT/rank_rehearsal_all.json 2> $OUT/rank_rehearsal_all.err | grep -v "$F" > $OUT/rank_rehearsal_all.txt; echo "rehearsal rc=$?"
$T python3 $R/tools/rank_rehearsal_all.py --gather-stream main 2> /dev/null | grep -v "$F" > $OUT/rank_rehearsal_all_gather_stream_main.txt
$T python3 $R/tools/placement_probe.py --contexts 6 --rounds 7 2>&1 | grep -v "$F" > $OUT/placement_probe_refresh.txt
echo "lines and rehearsals done"
cd /tmp
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
B="--direct --no-cpu-baseline --no-indexed --no-rehearsal --no-stream-record --no-box"
T="timeout -k 10 300"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 $B > $OUT/stats/bench.json 2> $OUT/stats/err.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_s1 -- python3 $R/bench.py --steps 20 --warmup 3 $B --streams 1 > $OUT/stats_s1/bench.json 2> $OUT/stats_s1/err.log
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/req_rd -- python3 $R/bench.py --steps 3 --warmup 1 $B --streams 1 > $OUT/req_rd/bench.json 2> $OUT/req_rd/err.log
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/req_wr -- python3 $R/bench.py --steps 3 --warmup 1 $B --streams 1 > $OUT/req_wr/bench.json 2> $OUT/req_wr/err.log
echo "soup counters done"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream -- python3 $R/bench.py --direct --config stream2048 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stream/bench.json 2> $OUT/stream/err.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/indexed -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 9 > $OUT/indexed/ab.log 2> $OUT/indexed/err.log
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/indexed_rd -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_rd/err.log
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/indexed_wr -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_wr/err.log
echo "indexed counters done"
cd $R
[ -f $R/tools/_ab/libvtmc_phases.so ] && VTMC_LIB=$R/tools/_ab/libvtmc_phases.so $T python3 $R/tools/emit_phases.py base indexed=1 2>&1 | grep -v "$F" > $OUT/emit_phases.txt
[ -f $R/tools/_ab/libvtmc_r05.so ] && $T python3 $R/tools/ab_two_libs.py r05=tools/_ab/libvtmc_r05.so r06=volumetricterrain_amd/libvtmc.so -- base indexed=1 --rounds 11 2>&1 | grep -v "$F" > $OUT/ab_r05_r06.txt
[ -f $R/tools/_ab/libvtmc_r05.so ] && $T python3 $R/tools/ab_two_libs.py r06=volumetricterrain_amd/libvtmc.so r05=tools/_ab/libvtmc_r05.so -- base indexed=1 --rounds 11 2>&1 | grep -v "$F" > $OUT/ab_r06_r05.txt
$T python3 $R/tools/dropin_route_bench.py > $OUT/dropin_route.txt 2>&1
bash $R/tools/pmc_sq.sh ${TAG}_final base > $OUT/sq_stdout.txt 2>&1
cp $R/gpurun_out/sq_${TAG}_final/summary.txt $OUT/sq_counters_soup.txt 2>/dev/null
rm -rf $OUT/*/runc/*.db $R/gpurun_out/sq_${TAG}_final/pass*/*/*.db 2>/dev/null
echo "profiles in $OUT"
