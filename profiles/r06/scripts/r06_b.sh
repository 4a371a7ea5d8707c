#!/bin/bash
# Round 6, second GPU call: exit probes in full (r6, r5, late), the hung configuration of round 5 through host_selftest, the tests the first call did
# not reach, the rehearsal with the collective on the steps' own streams, and the indexed emit kernel's phases + SQ counters.
TAG=${1:-r06b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/selftest
cd $R
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
T="timeout -k 10 420"
timeout -k 10 600 python -m pytest tests/test_lifecycle.py tests/test_bench_modes.py tests/test_own_queue_cpp_host.py tests/test_host_mirror.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
$T python3 $R/tools/rank_rehearsal_all.py --gather-stream main --json $OUT/rank_rehearsal_main.json 2> $OUT/rank_rehearsal_main.err | grep -v "$F" > $OUT/rank_rehearsal_main.txt; echo "rehearsal(main) rc=$?"
$T $R/tools/calib/mix2 4 box > $OUT/mix2_box.json 2>&1; cat $OUT/mix2_box.json
VTMC_LIB=$R/tools/_ab/libvtmc_phases.so $T python3 $R/tools/emit_phases.py indexed=1 "indexed=1,emit_ablate=1" base 2>&1 | grep -v "$F" > $OUT/emit_phases_indexed.txt; echo "phases rc=$?"
$T python3 $R/tools/ab_bench.py base indexed=1 --rounds 9 2>&1 | grep -v "$F" > $OUT/ab_indexed_baseline.txt
export TMPDIR=/tmp
( cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3"; do
    i=$((i + 1)); mkdir -p $OUT/sq_indexed/pass$i
    timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/sq_indexed/pass$i -- python3 $R/tools/ab_bench.py indexed=1 --rounds 2 > $OUT/sq_indexed/pass$i/out.log 2> $OUT/sq_indexed/pass$i/err.log || echo "pass $i failed"
done
python3 $R/tools/pmc_sq_summary.py $OUT/sq_indexed > $OUT/sq_counters_indexed.txt; rm -rf $OUT/sq_indexed/pass*/*/*.db 2>/dev/null )
echo "sq done"
# exit probes: a step that TIMES OUT (124 / 137) ends the chain; any other exit code is recorded and the chain goes on
P=$R/tools/calib/cumask_exit_probe
H=$R/host/_build/host_selftest
step() {   # step <label> <env...> -- cmd...
    local label=$1; shift
    echo "--- $label"
    env "$@" ; local rc=$?
    echo "--- $label: rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMED OUT: no further probe in this call"; exit 0; fi
}
{
  X="timeout -k 5 90"
  step "probe r6 (sync, destroy the CU-mask stream, then events and memory)" $X $P r6
  step "probe late (CU-mask stream destroyed by an atexit handler)" $X $P late
  step "host_selftest --gpu, main stream on its own queue, streams parked + destroyed at exit" VTMC_TEST_MAIN_STREAM_OWN_QUEUE=1 $X $H --gpu $OUT/selftest
  step "host_selftest --gpu, main stream on its own queue, streams destroyed in vtmc_destroy (round 6's order)" VTMC_TEST_MAIN_STREAM_OWN_QUEUE=1 VTMC_STREAM_POOL=0 $X $H --gpu $OUT/selftest
  step "probe r5 (round 5's order: memory and events first, the CU-mask stream last)" $X $P r5
  step "probe keep (CU-mask stream never destroyed)" $X $P keep
} > $OUT/exit_hang_probes.txt 2>&1
grep -- "---\|TIMED\|PROBE\|HOST-GPU" $OUT/exit_hang_probes.txt
echo "profiles in $OUT"
