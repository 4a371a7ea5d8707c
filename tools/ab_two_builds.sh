#!/bin/bash
# Same-box A/B of two library BUILDS (kernel changes that are no tuning key): the tree's library against tools/_ab/libvtmc_prev.so
# (build the older commit, copy its libvtmc.so there), alternating processes on one box.
# usage: gpurun -- 'bash tools/ab_two_builds.sh [variants ...] [-- extra ab_bench flags]'     default variants: "indexed=1" "base"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VARS=()
EXTRA=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; EXTRA=("$@"); break; fi
  VARS+=("$1"); shift
done
[ ${#VARS[@]} -eq 0 ] && VARS=("indexed=1" "base")
for i in 1 2 3; do
  echo "--- new"; timeout -k 10 200 python tools/ab_bench.py "${VARS[@]}" --rounds 7 "${EXTRA[@]}" 2>&1 | tail -${#VARS[@]}
  echo "--- prev"; VTMC_LIB=$R/tools/_ab/libvtmc_prev.so timeout -k 10 200 python tools/ab_bench.py "${VARS[@]}" --rounds 7 "${EXTRA[@]}" 2>&1 | tail -${#VARS[@]}
done
