#!/bin/bash
# Same-box A/B of two library BUILDS (kernel changes that are no tuning key): the tree's library against tools/_ab/libvtmc_prev.so
# (build the older commit, copy its libvtmc.so there), alternating processes on one box.  usage: gpurun -- 'bash tools/ab_two_builds.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2; do
  echo "--- new"; timeout -k 10 200 python tools/ab_bench.py "indexed=1" "base" --rounds 7 2>&1 | tail -2
  echo "--- prev"; VTMC_LIB=$R/tools/_ab/libvtmc_prev.so timeout -k 10 200 python tools/ab_bench.py "indexed=1" "base" --rounds 7 2>&1 | tail -2
done
