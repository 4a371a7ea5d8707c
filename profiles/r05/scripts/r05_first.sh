#!/bin/bash
# round 5, first lease: the GPU suite on the cleaned library, a bench line, the ablation A/B through the diagnostic build
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_a
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.json
VTMC_LIB=$R/tools/_ab/libvtmc_diag.so timeout -k 10 300 python tools/ab_bench.py base emit_ablate=1 emit_ablate=5 emit_once=0 "emit_once=0,emit_ablate=1" "emit_once=0,emit_ablate=5" --rounds 7 > $OUT/ab_ablation.txt 2>&1
cat $OUT/ab_ablation.txt
