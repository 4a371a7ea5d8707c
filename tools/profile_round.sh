#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + the two PMC passes the microarch
# guide prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass), all on bench.py itself.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write}
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/stats/bench.json 2> $OUT/stats/err.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch/bench.json 2> $OUT/fetch/err.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write/bench.json 2> $OUT/write/err.log
echo "profiles in $OUT"
