#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (made by tools/profile_round.sh) into profiles/<tag>/: kernel_stats.csv
(rocprofv3 --stats of bench.py), kernel_stats_stream2048.csv, pmc_fetch_write.json (per-kernel FETCH_SIZE /
WRITE_SIZE of the last dispatch, KiB as reported), fetch_calibration.json (the 40-byte-row gather: counter vs
known bytes), the un-profiled bench lines, and profiles/pmc_traffic.json (HBM bytes per launch: WRITE_SIZE +
FETCH_SIZE doubled -- the gfx950 correction of MI355X_MICROARCH.md for wide streaming reads; for the emit
kernel's 40-byte rows that doubling is an upper bound, see fetch_calibration.json)."""
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


def first(pattern):
    """The newest match: gpurun merges a re-run of the same tag into the files of the earlier one."""
    g = sorted(glob.glob(pattern), key=os.path.getmtime)
    return g[-1] if g else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(first(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), os.path.join(dst, "kernel_stats.csv"))
    shutil.copy(os.path.join(src, "stats", "bench.json"), os.path.join(dst, "bench_under_rocprof.json"))
    st = first(os.path.join(src, "stream", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st, os.path.join(dst, "kernel_stats_stream2048.csv"))
    # per-kernel averages over the REAL calls: the first extract's emit launch returns at once (buffer too small, finish() grows it
    # and launches again) and pulls the stats file's plain average down
    tr = first(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
    if tr:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(tr)):
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        bench_line = json.loads([ln for ln in open(os.path.join(src, "stats", "bench.json")) if ln.startswith("{")][-1])
        rows = {}
        for k, v in dur.items():
            if k.startswith("vtmc::"):
                real = [x for x in v if x >= 10000] or v
                rows[k] = {"calls": len(v), "calls_under_10us": len(v) - len(real), "avg_ms_real_calls": round(sum(real) / len(real) / 1e6, 4)}
        json.dump({"rocprofv3_kernel_trace": rows, "bench_hip_events_same_run": bench_line.get("kernels"),
                   "note": "same process: bench.py under rocprofv3 --kernel-trace --stats; the two averages of a kernel must agree"},
                  open(os.path.join(dst, "kernel_avg_vs_bench_events.json"), "w"), indent=1)
    for name in ("bench_n1.json", "bench_stream2048.json", "bench_2rank_one_device_gloo.json", "rank_step.txt", "rank_step_comm.txt"):
        f = os.path.join(src, name)
        if os.path.exists(f) and os.path.getsize(f):
            if name.endswith(".json"):   # torchrun's ranks also print connection chatter on stdout: keep the JSON line
                lines = [ln for ln in open(f) if ln.startswith("{")]
                open(os.path.join(dst, name), "w").write(lines[-1] if lines else "")
            else:
                shutil.copy(f, os.path.join(dst, name))
    fetch = last_per_kernel(first(os.path.join(src, "fetch", "*", "*_counter_collection.csv")), "FETCH_SIZE")
    write = last_per_kernel(first(os.path.join(src, "write", "*", "*_counter_collection.csv")), "WRITE_SIZE")
    json.dump({"FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write,
               "note": "last dispatch of each kernel in a bench.py run; FETCH_SIZE raw = TCC_EA0_RDREQ x 64 B"},
              open(os.path.join(dst, "pmc_fetch_write.json"), "w"), indent=1)
    ist = first(os.path.join(src, "indexed", "*", "*_kernel_stats.csv"))
    if ist:
        shutil.copy(ist, os.path.join(dst, "kernel_stats_indexed.csv"))
        try:
            fi = last_per_kernel(first(os.path.join(src, "indexed_fetch", "*", "*_counter_collection.csv")), "FETCH_SIZE")
            wi = last_per_kernel(first(os.path.join(src, "indexed_write", "*", "*_counter_collection.csv")), "WRITE_SIZE")
            json.dump({"FETCH_SIZE_KiB_raw": fi, "WRITE_SIZE_KiB": wi, "note": "tools/ab_bench.py indexed=1: the <true, true> emit kernel and the <true> classify kernel are the indexed pipeline's"},
                      open(os.path.join(dst, "pmc_fetch_write_indexed.json"), "w"), indent=1)
        except Exception as e:   # noqa: BLE001
            print("no indexed pmc:", e)
    calib = {}
    try:
        rows = json.load(open(os.path.join(src, "calib", "rows.json")))
        mix = json.load(open(os.path.join(src, "calib", "mix.json")))
        cf = last_per_kernel(first(os.path.join(src, "calib", "pmc", "*", "*_counter_collection.csv")), "FETCH_SIZE")
        raw = list(cf.values())[-1] * 1024.0
        calib = {"rows_kernel": rows, "mix_kernel": mix, "FETCH_SIZE_bytes_raw": raw,
                 "raw_over_known_row_bytes": round(raw / rows["known_bytes"], 4),
                 "doubled_would_imply_GBps": round(2 * raw / (rows["ms"] * 1e-3) / 1e9, 1),
                 "reading": "40-byte rows, each read once from a 4.5 GB buffer: the raw counter is ~1.23 x the row bytes; doubling it (the "
                            "guide's correction for 16-byte-per-lane streams) would mean more bytes per second than the chip can read, so for "
                            "this access width the true fetch traffic lies between the raw counter and twice it"}
        json.dump(calib, open(os.path.join(dst, "fetch_calibration.json"), "w"), indent=1)
    except Exception as e:   # noqa: BLE001
        print("no calibration:", e)
    traffic = {}
    for k in fetch:
        if k.startswith("vtmc::"):
            name = k.split("::")[1].split("<")[0]
            traffic[name + "_hbm_bytes"] = int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024)
            traffic[name + "_fetch_bytes_raw"] = int(fetch[k] * 1024)
            traffic[name + "_write_bytes"] = int(write.get(k, 0.0) * 1024)
    traffic["source"] = ("profiles/%s/pmc_fetch_write.json; *_hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction for wide reads; an upper "
                         "bound for the emit kernel's 40-byte rows, profiles/%s/fetch_calibration.json)" % (tag, tag))
    if calib:
        traffic["mix_stream_ceiling_GBps"] = calib["mix_kernel"]["GBps_total"]
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "kernel_stats.csv")).read())
    print(json.dumps(traffic, indent=1))
    print(json.dumps(calib, indent=1))


if __name__ == "__main__":
    main()
