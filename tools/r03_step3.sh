#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_step3
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout -k 10 300 python tools/ab_bench.py "emit_cell_masks=0" "emit_cell_masks=1" "emit_cell_masks=0,emit_once=0" "emit_cell_masks=1,emit_once=0" "emit_cell_masks=0,indexed=1" "emit_cell_masks=1,indexed=1" --rounds 7 > $OUT/ab_masks.txt 2>&1
cat $OUT/ab_masks.txt
