#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_d
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_tuning_matrix.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "not config_1024 and not max_size and not streaming" > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
timeout -k 10 300 python tools/ab_bench.py base classify_wide=1 "classify_wide=1,classify_wgs_per_cu=0" "classify_wide=1,classify_wgs_per_cu=2" "indexed=1" "indexed=1,classify_wide=1" --rounds 9 > $OUT/ab_classify_wide.txt 2>&1
cat $OUT/ab_classify_wide.txt
VTMC_LIB=$R/tools/_ab/libvtmc_diag.so timeout -k 10 300 python tools/ab_bench.py base emit_ablate=64 emit_ablate=1 "emit_once=0" "emit_once=0,emit_ablate=64" "emit_once=0,emit_ablate=1" --rounds 7 > $OUT/ab_store_window.txt 2>&1
cat $OUT/ab_store_window.txt
