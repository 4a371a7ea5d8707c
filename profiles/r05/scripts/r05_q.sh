#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_q
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_bench_modes.py tests/test_gpu_parity.py tests/test_lifecycle.py tests/test_host_mirror.py -m gpu -x -q -p no:cacheprovider -k "bench or rccl or lifecycle or host_mirror or stream" > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -4 $OUT/pytest_subset.log
timeout -k 10 300 python tools/rank_overlap_probe.py 8 --rounds 3 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/rank_overlap_probe.txt
cat $OUT/rank_overlap_probe.txt
for q in "" "--stream-two-queues" "--stream-two-queues --sampler-wgs 3" "--sampler-wgs 3"; do
  echo "=== stream2048 $q" >> $OUT/stream2048_two_queues.txt
  timeout -k 10 300 python bench.py --config stream2048 --no-cpu-baseline $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['triangles_total'], d['kernels_ms_per_step_serialised'])" >> $OUT/stream2048_two_queues.txt
done
cat $OUT/stream2048_two_queues.txt
VTMC_BENCH_FORCE_COMM=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-indexed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('world-of-one comm, default:', d['ms_per_step'], d['one_stream_ms_per_step'], d['allgather_ms'])"
VTMC_BENCH_FORCE_COMM=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-indexed --gather-stream side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('world-of-one comm, side:', d['ms_per_step'], d['one_stream_ms_per_step'], d['allgather_ms'])"
