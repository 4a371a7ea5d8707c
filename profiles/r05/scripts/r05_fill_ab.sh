#!/bin/bash
# the sampler alone (one 64-chunk batch of the 2048^3 fbm8 world, sign words kept), library against library on one box, alternating
# usage: r05_fill_ab.sh lib [lib ...]
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib: $(VTMC_FILL_SIGNS=1 VTMC_LIB=$lib python tools/fill_only.py fbm8 2>/dev/null)"
  done
done
