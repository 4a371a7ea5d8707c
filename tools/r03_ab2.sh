#!/bin/bash
# same-box A/B of two library builds: the tree's (new) against tools/_ab/libvtmc_prev.so, alternating processes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2; do
  echo "--- new"; timeout -k 10 200 python tools/ab_bench.py "indexed=1" "base" --rounds 7 2>&1 | tail -2
  echo "--- prev"; VTMC_LIB=$R/tools/_ab/libvtmc_prev.so timeout -k 10 200 python tools/ab_bench.py "indexed=1" "base" --rounds 7 2>&1 | tail -2
done
