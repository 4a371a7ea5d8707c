#!/bin/bash
# the request-by-size PMC passes again (profiles/pmc_traffic.json is tied to the SHA-256 of the kernel sources; a diagnostic line went into
# emit_kernels.hip after the round's refresh) -- the same four commands as tools/profile_round5.sh, into the same directories
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r05
mkdir -p $OUT/req_rd $OUT/req_wr $OUT/indexed_rd $OUT/indexed_wr
rm -rf $OUT/req_rd/* $OUT/req_wr/* $OUT/indexed_rd/* $OUT/indexed_wr/*
export TMPDIR=/tmp
cd /tmp
T="timeout -k 10 300"
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/req_rd -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed --streams 1 > $OUT/req_rd/bench.json 2> $OUT/req_rd/err.log || exit 1
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/req_wr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-indexed --streams 1 > $OUT/req_wr/bench.json 2> $OUT/req_wr/err.log || exit 1
$T rocprofv3 --pmc $RD --kernel-trace --output-format csv -d $OUT/indexed_rd -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_rd/err.log || exit 1
$T rocprofv3 --pmc $WR --kernel-trace --output-format csv -d $OUT/indexed_wr -- python3 $R/tools/ab_bench.py "indexed=1" --rounds 2 > /dev/null 2> $OUT/indexed_wr/err.log || exit 1
echo "pmc passes done"
