#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_step2
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout -k 10 200 python tools/ab_bench.py "emit_once=0" "emit_once=1" "emit_once=1,emit_wgs_per_cu=2" "emit_once=1,emit_ablate=1" "emit_once=1,emit_ablate=4" "emit_once=1,emit_ablate=5" --rounds 7 > $OUT/ab_once.txt 2>&1
cat $OUT/ab_once.txt
