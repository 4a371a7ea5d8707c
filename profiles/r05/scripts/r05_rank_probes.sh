#!/bin/bash
# the two rank-step rehearsals of tools/profile_round5.sh again (the probe used to die in the interpreter's tear-down AFTER printing its table:
# torch events / pinned tensors outliving the contexts' streams; fixed in the tool), exit codes shown
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r05; mkdir -p $OUT
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout -k 10 300 python3 $R/tools/rank_overlap_probe.py 8 --rounds 3 > $OUT/rop8.raw 2>&1; echo "rank_overlap_probe 8: rc=$?"
grep -v "$F" $OUT/rop8.raw > $OUT/rank_overlap_probe.txt
timeout -k 10 300 python3 $R/tools/rank_overlap_probe.py 1 --rounds 2 > $OUT/rop1.raw 2>&1; echo "rank_overlap_probe 1: rc=$?"
grep -v "$F" $OUT/rop1.raw > $OUT/rank_overlap_probe_w1.txt
timeout -k 10 300 python3 $R/tools/depth_probe.py 8 --rounds 3 --steps 200 --comm > $OUT/depth8.raw 2>&1; echo "depth_probe 8 --comm: rc=$?"
tail -3 $OUT/rank_overlap_probe.txt
