#!/bin/bash
# Instruction-count pass only (SQ_INSTS_*) over tools/ab_bench.py variants: gpurun_out/insts_<tag>/<i>/
# usage: tools/pmc_insts.sh <tag> variant [variant ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/insts_$TAG
export TMPDIR=/tmp
cd /tmp
i=0
for V in "$@"; do
    i=$((i + 1))
    mkdir -p $OUT/$i
    timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/$i -- python3 $R/tools/ab_bench.py "$V" --rounds 2 > $OUT/$i/out.log 2> $OUT/$i/err.log
    echo "== $V"
    python3 - $OUT/$i <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "emit_kernel" in k:
        per[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(per.items()):
    print("  %-34s %-18s %14.0f" % (k[-34:], c, v[-1]))
PY
done
