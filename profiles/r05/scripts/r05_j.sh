#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_j
mkdir -p $OUT
cd $R
for t in "classify_wgs_per_cu=3" "classify_wgs_per_cu=0" "classify_wgs_per_cu=4" "classify_wgs_per_cu=2" "emit_wgs_per_cu=2,classify_wgs_per_cu=0" "emit_wgs_per_cu=2,classify_wgs_per_cu=4" "emit_once=0" "emit_once=0,classify_wgs_per_cu=0"; do
  timeout -k 10 200 python tools/rank_overlap_probe.py 1 --rounds 2 --only two --tune $t 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" >> $OUT/overlap_tuning_w1.txt
done
cat $OUT/overlap_tuning_w1.txt
