#!/usr/bin/env python3
"""Per-kernel means of the counters tools/pmc_sq.sh collected (last two dispatches of each kernel)."""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    table = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(root, "pass*", "*", "*_counter_collection.csv"))):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in per.items():
            v = v[-2:]
            table.setdefault(k, collections.OrderedDict())[c] = sum(v) / len(v)
    for k, cs in table.items():
        if "vtmc" not in k:
            continue
        print(k)
        for c, v in cs.items():
            print("    %-36s %16.0f" % (c, v))


if __name__ == "__main__":
    main()
