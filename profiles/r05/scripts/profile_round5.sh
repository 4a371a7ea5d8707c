#!/bin/bash
# Round 5's refresh on the GPU box (through gpurun); tools/summarize_profiles.py r05 distils it into profiles/r05/.
#   pytest_gpu.log            the full GPU suite on the library of this snapshot
#   bench_n1.json             the driver's command (two contexts, a stream -- hardware queue -- each; kernel rooflines from its one-stream region)
#   stats / stats_s1          rocprofv3 --kernel-trace --stats of that command and of `--streams 1`
#   req_rd / req_wr           memory-side requests BY SIZE (exact HBM-side bytes per launch), `--streams 1`
#   indexed*, stream          the indexed pipeline, config 5
#   emit_phases.txt, ab_*     diagnostic builds (tools/_ab/libvtmc_phases.so, libvtmc_diag.so: tools/build_diagnostics.py beforehand)
#   rank_overlap_probe*.txt   one rank of eight / the whole world, every way of driving two steps in flight
# usage: tools/profile_round5.sh [tag]      -> gpurun_out/prof_<tag>/
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/stats_s1 $OUT/req_rd $OUT/req_wr $OUT/stream $OUT/indexed $OUT/indexed_rd $OUT/indexed_wr
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
export TMPDIR=/tmp
cd /tmp
T="timeout -k 10 300"
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
$T python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done"
$T python3 $R/bench.py --config stream2048 > $OUT/bench_stream2048.json 2> $OUT/bench_stream.err
VTMC_BENCH_ONE_DEVICE=1 VTMC_BENCH_BACKEND=gloo $T python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29719 $R/bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_2rank_one_device_gloo.json 2> $OUT/bench_2rank.err
VTMC_BENCH_FORCE_COMM=1 $T python3 $R/bench.py --no-cpu-baseline --no-indexed > $OUT/bench_world_of_one_comm.json 2> $OUT/bench_comm.err
$T python3 $R/tools/rank_overlap_probe.py 8 --rounds 3 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/rank_overlap_probe.txt
$T python3 $R/tools/rank_overlap_probe.py 1 --rounds 2 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/rank_overlap_probe_w1.txt
echo "rehearsals done"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-indexed > $OUT/stats/bench.json 2> $OUT/stats/err.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_s1 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-indexed --streams 1 > $OUT/stats_s1/bench.json 2> $OUT/stats_s1/err.log
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/req_rd -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed --streams 1 > $OUT/req_rd/bench.json 2> $OUT/req_rd/err.log
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/req_wr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed --streams 1 > $OUT/req_wr/bench.json 2> $OUT/req_wr/err.log
echo "soup counters done"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream -- python3 $R/bench.py --config stream2048 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stream/bench.json 2> $OUT/stream/err.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/indexed -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 9 > $OUT/indexed/ab.log 2> $OUT/indexed/err.log
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/indexed_rd -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_rd/err.log
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/indexed_wr -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_wr/err.log
echo "indexed counters done"
cd $R
[ -f $R/tools/_ab/libvtmc_phases.so ] && VTMC_LIB=$R/tools/_ab/libvtmc_phases.so $T python3 $R/tools/emit_phases.py base emit_ablate=1 emit_ablate=5 > $OUT/emit_phases.txt 2>&1
[ -f $R/tools/_ab/libvtmc_diag.so ] && VTMC_LIB=$R/tools/_ab/libvtmc_diag.so $T python3 $R/tools/ab_bench.py base emit_ablate=64 emit_ablate=1 emit_ablate=5 emit_once=0 "emit_once=0,emit_ablate=64" "emit_once=0,emit_ablate=1" "emit_once=0,emit_ablate=5" --rounds 7 > $OUT/ab_emit_ablation.txt 2>&1
[ -f $R/tools/_ab/libvtmc_prev.so ] && $T python3 $R/tools/ab_two_libs.py r05=volumetricterrain_amd/libvtmc.so r05_first_rework=tools/_ab/libvtmc_r05c1.so r04=tools/_ab/libvtmc_prev.so -- base emit_once=0 indexed=1 --rounds 11 > $OUT/ab_three_libs.txt 2>&1
$T python3 $R/tools/dropin_route_bench.py > $OUT/dropin_route.txt 2>&1
bash $R/tools/pmc_sq.sh ${TAG}_final base > $OUT/sq_stdout.txt 2>&1
cp $R/gpurun_out/sq_${TAG}_final/summary.txt $OUT/sq_counters_soup.txt 2>/dev/null
echo "profiles in $OUT"
