#!/usr/bin/env python3
"""Prints the kernel timeline (start offset, duration, gap to the previous kernel, in microseconds) of the
last N dispatches of a rocprofv3 --kernel-trace CSV."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vtmc::", "")[:40]
    print("%-42s start %9.2f  dur %8.2f  gap %7.2f" % (name, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
