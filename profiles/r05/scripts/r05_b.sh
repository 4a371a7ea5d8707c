#!/bin/bash
# parity of the reworked emit kernels (subset), then new vs prev build A/B, then phase shares
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_b
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_indexed.py tests/test_random_shapes.py tests/test_tuning_matrix.py -m gpu -x -q -p no:cacheprovider -k "not config_1024 and not max_size and not streaming" > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
bash tools/ab_two_builds.sh base indexed=1 emit_once=0 > $OUT/ab_new_vs_prev.txt 2>&1
cat $OUT/ab_new_vs_prev.txt
VTMC_LIB=$R/tools/_ab/libvtmc_phases.so timeout -k 10 240 python3 tools/emit_phases.py base emit_ablate=1 emit_ablate=5 > $OUT/emit_phases.txt 2>&1
cat $OUT/emit_phases.txt
