#!/bin/bash
# config 5: both contexts' extracts on one high-priority stream against the shipped driver (a context's sampler + extract on its own-queue stream)
OUT=gpurun_out/r05_prio; mkdir -p $OUT; : > $OUT/ab.txt
for rep in 1 2 3; do
  for q in "" "--stream-extract-priority"; do
    echo "=== rep $rep stream2048 $q" >> $OUT/ab.txt
    timeout -k 10 200 python bench.py --config stream2048 --no-cpu-baseline $q 2> $OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['triangles_total'], d['kernels_ms_per_step_serialised'])" >> $OUT/ab.txt || { tail -5 $OUT/err.log; exit 1; }
  done
done
cat $OUT/ab.txt
