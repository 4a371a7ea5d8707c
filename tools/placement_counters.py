#!/usr/bin/env python3
"""Per-DISPATCH memory-side counters of the emit kernel beside its duration, context by context (round 6: the emit kernel's time is a property of
the allocation its output buffer is -- which counter moves with it?).  Input: directories written by
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/placement_probe.py --contexts N --rounds R --no-realloc
The probe's emit dispatches go round the contexts in turn: the last N * R of them are the measured rounds, dispatch i belongs to context i % N.
    python tools/placement_counters.py <N> <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import statistics
import sys

n_ctx = int(sys.argv[1])
for d in sys.argv[2:]:
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not f:
        print("%s: no counter file" % d)
        continue
    rows = collections.OrderedDict()   # dispatch id -> {counter: value, "ns": duration}
    for r in csv.DictReader(open(f[-1])):
        if "emit_kernel" not in r["Kernel_Name"]:
            continue
        e = rows.setdefault(int(r["Dispatch_Id"]), {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [v for _, v in sorted(rows.items()) if v["ns"] > 100000]   # the first extract's emit launch returns at once (buffer too small)
    names = [k for k in disp[-1] if k != "ns"]
    rounds = (len(disp) // n_ctx) - 2
    use = disp[-n_ctx * rounds:] if rounds > 0 else disp
    print("%s: %d emit dispatches, the last %d taken as %d rounds over %d contexts" % (d, len(disp), len(use), rounds, n_ctx))
    print("  %-8s %10s  %s" % ("context", "ms (med)", "  ".join("%22s" % n.replace("_sum", "") for n in names)))
    for c in range(n_ctx):
        mine = use[c::n_ctx]
        print("  %-8d %10.4f  %s" % (c, statistics.median(m["ns"] for m in mine) / 1e6, "  ".join("%22.0f" % statistics.median(m.get(n, 0.0) for m in mine) for n in names)))
