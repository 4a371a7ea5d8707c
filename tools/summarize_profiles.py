#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (made by tools/profile_round.sh) into profiles/<tag>/:
kernel_stats.csv (rocprofv3 --stats), pmc_fetch_write.json (per-kernel FETCH_SIZE / WRITE_SIZE of the
last dispatch, KiB as reported) and profiles/pmc_traffic.json (HBM bytes per launch with the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests as 64 -> doubled)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_per_kernel(path, counter):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"].split("(")[0].replace("void ", "")] = float(r["Counter_Value"])
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
    shutil.copy(os.path.join(src, "stats", "bench.json"), os.path.join(dst, "bench_under_rocprof.json"))
    fetch = last_per_kernel(glob.glob(os.path.join(src, "fetch", "*", "*_counter_collection.csv"))[0], "FETCH_SIZE")
    write = last_per_kernel(glob.glob(os.path.join(src, "write", "*", "*_counter_collection.csv"))[0], "WRITE_SIZE")
    json.dump({"FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write,
               "note": "last dispatch of each kernel in a bench.py run; FETCH_SIZE raw = TCC_EA0_RDREQ x 64 B"},
              open(os.path.join(dst, "pmc_fetch_write.json"), "w"), indent=1)
    traffic = {}
    for k in fetch:
        if k.startswith("vtmc::"):
            name = k.split("::")[1].split("<")[0]
            traffic[name + "_hbm_bytes"] = int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024)
    traffic["source"] = "profiles/%s/pmc_fetch_write.json; bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB->B), gfx950 correction" % tag
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "kernel_stats.csv")).read())
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
