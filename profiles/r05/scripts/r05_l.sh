#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_l
mkdir -p $OUT
cd $R
for v in cumask cumask_one ctx_own plain; do
  timeout -k 10 120 python tools/stream_overlap_variants.py $v 2>&1 | grep -v "amdgpu.ids" >> $OUT/stream_variants2.txt
done
GPU_MAX_HW_QUEUES=4 timeout -k 10 120 python tools/stream_overlap_variants.py plain 2>&1 | grep -v amdgpu.ids >> $OUT/stream_variants2.txt
GPU_MAX_HW_QUEUES=16 timeout -k 10 120 python tools/stream_overlap_variants.py plain 2>&1 | grep -v amdgpu.ids >> $OUT/stream_variants2.txt
cat $OUT/stream_variants2.txt
