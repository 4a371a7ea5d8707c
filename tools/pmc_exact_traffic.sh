#!/bin/bash
# Exact memory-side request sizes of the kernels (settles what FETCH_SIZE means on the emit kernel's 40-byte rows: FETCH_SIZE is
# TCC_EA0_RDREQ x 64 B, right only if every request is 64 bytes long): requests by size, reads and writes, one rocprofv3 run per
# counter group (counters only with --kernel-trace, as the pool requires).
# usage: tools/pmc_exact_traffic.sh <tag> [ab_bench variant ...]   -> gpurun_out/traffic_<tag>/<n>/summary.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
PA="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
PB="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
n=0
[ $# -eq 0 ] && set -- "base"
for VAR in "$@"; do
    n=$((n + 1))
    OUT=$R/gpurun_out/traffic_$TAG/$n
    mkdir -p $OUT
    echo "$VAR" > $OUT/variant.txt
    i=0
    for P in "$PA" "$PB"; do
        i=$((i + 1))
        mkdir -p $OUT/pass$i
        timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $R/tools/ab_bench.py "$VAR" --rounds 2 > $OUT/pass$i/out.log 2> $OUT/pass$i/err.log || echo "pass $i failed"
    done
    python3 $R/tools/pmc_sq_summary.py $OUT > $OUT/summary.txt
    echo "== $VAR"; cat $OUT/summary.txt
    rm -rf $OUT/pass*/*/
done
