#!/bin/bash
# Round 6's refresh, part 3: the full GPU suite on the FINAL library, the same suite's parity core through the poisoned-LDS build, the emit phases.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT; cd $R
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
VTMC_LIB=$R/tools/_ab/libvtmc_poison.so timeout -k 10 600 python -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_indexed.py tests/test_random_shapes.py tests/test_tuning_matrix.py tests/test_terrain.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_poisoned_lds.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_poisoned_lds.log; tail -3 $OUT/pytest_gpu_poisoned_lds.log
VTMC_LIB=$R/tools/_ab/libvtmc_phases.so timeout -k 10 300 python3 $R/tools/emit_phases.py base indexed=1 2>&1 | grep -v "$F" > $OUT/emit_phases.txt; cat $OUT/emit_phases.txt
timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_n1_driver_command.json 2> $OUT/bench_n1_driver_command.err; echo "bench rc=$?"
