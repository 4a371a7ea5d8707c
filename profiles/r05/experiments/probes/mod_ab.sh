#!/bin/bash
# cache-policy bits on the emit kernels' tile loads: time (ab_bench) and memory-side read requests by size, per library build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for L in "" nt sc1 sc0_sc1 sc0_sc1_nt; do
  if [ -n "$L" ]; then export VTMC_LIB=$R/tools/_ab/libvtmc_mod_$L.so; else unset VTMC_LIB; fi
  echo "=== tile loads: ${L:-default}"
  timeout -k 10 100 python tools/ab_bench.py base "indexed=1" --rounds 7 2>&1 | tail -2
  bash tools/pmc_exact_traffic.sh mod_${L:-default} base > /dev/null 2>&1
  grep -A4 "emit_kernel" gpurun_out/traffic_mod_${L:-default}/1/summary.txt | head -5
done
