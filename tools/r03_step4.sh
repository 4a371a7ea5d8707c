#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_step4
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout -k 10 300 python tools/ab_bench.py "indexed=1" "indexed=1,emit_ablate=16" "indexed=1,emit_ablate=32" "indexed=1,emit_ablate=64" "indexed=1,emit_wgs_per_cu=3" "base" --rounds 7 > $OUT/ab_indexed.txt 2>&1
cat $OUT/ab_indexed.txt
VTMC_LIB=$R/tools/_ab/libvtmc_r02.so timeout -k 10 200 python tools/ab_bench.py "indexed=1" --rounds 7 > $OUT/ab_r02lib.txt 2>&1
cat $OUT/ab_r02lib.txt
