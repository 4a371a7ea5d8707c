#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_c
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -i -c "sq_" $OUT/counters_avail.txt
cd $R
bash tools/pmc_sq.sh r05c base > $OUT/sq_summary_stdout.txt 2>&1
tail -3 $OUT/sq_summary_stdout.txt
