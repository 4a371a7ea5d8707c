#!/bin/bash
# Round 6, third GPU call: parity + A/B of the z-marching classify kernel, then the exit probe by what rides on the stream.
TAG=${1:-r06c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
T="timeout -k 10 420"
timeout -k 10 600 python -m pytest tests/test_tuning_matrix.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_tuning.log 2>&1
rc=$?; echo "pytest rc=$rc" >> $OUT/pytest_tuning.log; tail -3 $OUT/pytest_tuning.log
if [ $rc -eq 0 ]; then
  $T python3 $R/tools/ab_bench.py base classify_column=2 classify_column=4 classify_column=8 classify_column=16 --rounds 7 2>&1 | grep -v "$F" > $OUT/ab_classify_column.txt
  $T python3 $R/tools/ab_bench.py base "classify_column=4,classify_wgs_per_cu=0" "classify_column=4,classify_wgs_per_cu=2" "classify_column=4,classify_wgs_per_cu=4" "classify_column=4,classify_wgs_per_cu=5" "classify_column=4,classify_wgs_per_cu=6" "classify_column=8,classify_wgs_per_cu=0" "classify_column=8,classify_wgs_per_cu=4" "classify_column=8,classify_wgs_per_cu=6" --rounds 7 2>&1 | grep -v "$F" >> $OUT/ab_classify_column.txt
  cat $OUT/ab_classify_column.txt
fi
P=$R/tools/calib/cumask_exit_probe
step() {
    local label=$1; shift
    echo "--- $label"
    env "$@" ; local rc=$?
    echo "--- $label: rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: no further probe in this call"; exit 0; fi
}
{
  X="timeout -k 5 60"
  step "r6, kernels and events only on the CU-mask stream" $X $P r6 ""
  step "r6, + a kernel that writes mapped pinned memory" $X $P r6 w
  step "r6, + the pinned host-to-device copy" $X $P r6 h
  step "r6, + the device-to-host copy into pinned memory" $X $P r6 d
  step "r6, + the device-to-host copy into pageable memory" $X $P r6 p
  step "plain stream, everything" $X $P plain hdpw
} > $OUT/exit_probe_by_operation.txt 2>&1
grep -- "--- .*rc=\|TIMED" $OUT/exit_probe_by_operation.txt
echo "profiles in $OUT"
