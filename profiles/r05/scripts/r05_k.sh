#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_k
mkdir -p $OUT
cd $R
for v in plain set_stream streams_first rccl_first priority many; do
  timeout -k 10 120 python tools/stream_overlap_variants.py $v 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" >> $OUT/stream_variants.txt
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python tools/stream_overlap_variants.py plain 2>&1 | grep -v amdgpu.ids >> $OUT/stream_variants.txt
GPU_MAX_HW_QUEUES=2 timeout -k 10 120 python tools/stream_overlap_variants.py plain 2>&1 | grep -v amdgpu.ids >> $OUT/stream_variants.txt
cat $OUT/stream_variants.txt
bash tools/r05_j.sh
