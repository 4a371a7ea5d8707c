#!/bin/bash
# Round 6's refresh, part 2 (a gpurun call lasts 20 minutes at most): the profiler passes -- rocprofv3 statistics and memory-side counters of the
# bench (--direct: the program itself behind `--`), config 5, the indexed pipeline, the diagnostic builds, SQ counters.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/stats_s1 $OUT/req_rd $OUT/req_wr $OUT/stream $OUT/indexed $OUT/indexed_rd $OUT/indexed_wr
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
export TMPDIR=/tmp
cd /tmp
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
B="--direct --no-cpu-baseline --no-indexed --no-rehearsal --no-stream-record --no-terrain-record --no-box"
T="timeout -k 10 150"
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
