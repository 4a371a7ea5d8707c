#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_h
mkdir -p $OUT
cd $R
timeout -k 10 300 python tools/rank_overlap_probe.py 8 --rounds 3 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/rank_overlap_probe.txt
cat $OUT/rank_overlap_probe.txt
timeout -k 10 300 python tools/rank_overlap_probe.py 1 --rounds 2 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/rank_overlap_probe_w1.txt
cat $OUT/rank_overlap_probe_w1.txt
