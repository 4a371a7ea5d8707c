#!/bin/bash
# round 5: the sampler's instruction trim -- parity, then the 2048^3 pass of this build beside HEAD's library (tools/_ab/libvtmc_head.so), alternating
OUT=gpurun_out/r05_s; mkdir -p $OUT
set -o pipefail
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "density or sampler or sign_bits or config5 or stream" > $OUT/pytest_sampler.log 2>&1 || { tail -30 $OUT/pytest_sampler.log; exit 1; }
tail -3 $OUT/pytest_sampler.log
: > $OUT/stream_ab.txt
for rep in 1 2 3; do
  for lib in tools/_ab/libvtmc_head.so volumetricterrain_amd/libvtmc.so; do
    echo "=== rep $rep $lib" >> $OUT/stream_ab.txt
    VTMC_LIB=$lib timeout -k 10 200 python bench.py --config stream2048 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['triangles_total'], d['kernels_ms_per_step_serialised'])" >> $OUT/stream_ab.txt || exit 1
  done
done
cat $OUT/stream_ab.txt
