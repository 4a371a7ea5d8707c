#!/bin/bash
# memory-side read requests of the emit kernel with the canonical and with the Morton list order (one rocprofv3 --pmc pass each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_morton; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in base emit_list_morton=1; do
  d=$OUT/$(echo $v | tr '=' '_'); mkdir -p $d
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $d -- python3 $R/tools/ab_bench.py "$v" --rounds 2 > $d/out.log 2> $d/err.log || echo "pass failed: $v"
  python3 - "$d" "$v" <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "emit_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v[-2:]) / len(v[-2:]) / 1e6, 3) for k, v in per.items()}, "(millions, mean of the last two launches)")
PY
done
