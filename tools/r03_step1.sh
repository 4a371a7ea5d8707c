#!/bin/bash
# round 3, step 1: parity of the asynchronous prefetch + same-box A/B against round 2's loop and round 2's library
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_step1
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout -k 10 200 python tools/ab_bench.py "emit_async=0" "emit_async=1" "emit_async=1,indexed=1" "emit_async=0,indexed=1" --rounds 9 > $OUT/ab_async.txt 2>&1
cat $OUT/ab_async.txt
VTMC_LIB=$R/tools/_ab/libvtmc_r02.so timeout -k 10 200 python tools/ab_bench.py "base" "indexed=1" --rounds 9 > $OUT/ab_r02lib.txt 2>&1
cat $OUT/ab_r02lib.txt
timeout -k 10 200 python tools/ab_bench.py "emit_async=1,emit_ablate=1" "emit_async=1,emit_ablate=4" "emit_async=1,emit_ablate=5" "emit_async=1,emit_ablate=7" "emit_async=1,emit_wgs_per_cu=3" "emit_async=1,emit_wgs_per_cu=2" --rounds 5 > $OUT/ab_ablate.txt 2>&1
cat $OUT/ab_ablate.txt
