#!/bin/bash
# Round 4's refresh, on the GPU box (through gpurun): tools/profile_round3.sh (bench lines, rocprofv3 stats, requests by size, ceilings, A/Bs,
# rank rehearsals) plus what this round added -- the plain one-float4-per-thread kernels and the fronts sweep of tools/calib/mix2, the emit
# kernel's phase shares and the classify timeline (diagnostic builds, tools/_ab/*.so built beforehand by tools/build_diagnostics.py), the full
# GPU suite's log.  tools/summarize_profiles.py <tag> distils it back home.
# usage: tools/profile_round4.sh <tag>      -> gpurun_out/prof_<tag>/
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/calib
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
bash tools/profile_round3.sh $TAG
T="timeout -k 10 240"
$T $R/tools/calib/mix2 8 copy > $OUT/calib/plain.jsonl 2>> $OUT/calib/err.log
$T $R/tools/calib/mix2 4 fronts > $OUT/calib/fronts.jsonl 2>> $OUT/calib/err.log
[ -f $R/tools/_ab/libvtmc_phases.so ] && VTMC_LIB=$R/tools/_ab/libvtmc_phases.so $T python3 $R/tools/emit_phases.py base emit_ablate=1 emit_ablate=5 > $OUT/emit_phases.txt 2>&1
[ -f $R/tools/_ab/libvtmc_timeline.so ] && VTMC_LIB=$R/tools/_ab/libvtmc_timeline.so $T python3 $R/tools/classify_timeline.py 512 64 > $OUT/classify_timeline.txt 2>&1
$T python3 $R/tools/dropin_route_bench.py > $OUT/dropin_route.txt 2>&1
echo "round-4 extras in $OUT"
