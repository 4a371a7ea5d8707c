#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (made by tools/profile_round3.sh) into profiles/<tag>/: kernel_stats.csv
(rocprofv3 --stats of bench.py), kernel_stats_stream2048.csv, pmc_requests_by_size.json (per-kernel memory-side
requests by size of the last dispatch: exact read / write bytes), memory_ceilings.json + the raw tools/calib/mix2
lines, the same-box A/B logs, the un-profiled bench lines, and profiles/pmc_traffic.json (HBM-side bytes per launch
with the SHA-256 of the kernel sources they were measured on: bench.py reports `traffic` only while that still matches)."""
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
        # round 5: bench.py's timed region has the two contexts on a stream (hardware queue) each and its kernels overlap; the kernel rooflines
        # come from its second, one-stream region.  A call of the trace belongs to that region when no kernel of another context's step was on
        # the chip beside it: the average over THOSE calls is what must agree with bench.py's HIP events (kernels.*.avg_ms).
        calls = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(tr))
                        if "vtmc::" in r["Kernel_Name"]), key=lambda c: c[0])
        alone = collections.defaultdict(list)
        for i, (s0, e0, k) in enumerate(calls):
            if e0 - s0 < 10000:
                continue
            beside = any(calls[j][0] < e0 - 2000 and calls[j][1] > s0 + 2000 and calls[j][1] - calls[j][0] >= 10000 for j in range(max(0, i - 6), min(len(calls), i + 7)) if j != i)
            if not beside:
                alone[k].append(e0 - s0)
        for k, v in alone.items():
            if k in rows:
                rows[k]["calls_alone_on_the_chip"] = len(v)
                rows[k]["avg_ms_calls_alone"] = round(sum(v) / len(v) / 1e6, 4)
        json.dump({"rocprofv3_kernel_trace": rows, "bench_hip_events_same_run": bench_line.get("kernels"),
                   "note": "same process: bench.py under rocprofv3 --kernel-trace --stats.  avg_ms_calls_alone (the calls of the one-stream region, where no other "
                           "kernel of the library shares the chip) must agree with bench.py's kernels.*.avg_ms; avg_ms_real_calls mixes them with the spans of the "
                           "two-queue region"},
                  open(os.path.join(dst, "kernel_avg_vs_bench_events.json"), "w"), indent=1)
    s1 = first(os.path.join(src, "stats_s1", "*", "*_kernel_stats.csv"))
    if s1:
        shutil.copy(s1, os.path.join(dst, "kernel_stats_one_stream.csv"))
        shutil.copy(os.path.join(src, "stats_s1", "bench.json"), os.path.join(dst, "bench_under_rocprof_one_stream.json"))
    for name in ("bench_n1.json", "bench_stream2048.json", "bench_2rank_one_device_gloo.json", "bench_world_of_one_comm.json", "rank_step.txt", "rank_step_comm.txt",
                 "rank_overlap_probe.txt", "rank_overlap_probe_w1.txt", "ab_three_libs.txt", "sq_counters_soup.txt", "rank_rehearsal_all.txt",
                 "rank_rehearsal_all_gather_stream_main.txt", "placement_probe_refresh.txt", "ab_r05_r06.txt", "ab_r06_r05.txt"):
        f = os.path.join(src, name)
        if os.path.exists(f) and os.path.getsize(f):
            if name.endswith(".json"):   # torchrun's ranks also print connection chatter on stdout: keep the JSON line
                lines = [ln for ln in open(f) if ln.startswith("{")]
                open(os.path.join(dst, name), "w").write(lines[-1] if lines else "")
            else:
                shutil.copy(f, os.path.join(dst, name))
    def sized_requests(rd_dir, wr_dir):
        """Exact memory-side bytes per launch from the requests BY SIZE (last dispatch of each kernel): reads 32 / 64 / 128 B, writes 32 / 64 B."""
        rd_csv = first(os.path.join(src, rd_dir, "*", "*_counter_collection.csv"))
        wr_csv = first(os.path.join(src, wr_dir, "*", "*_counter_collection.csv"))
        if not rd_csv or not wr_csv:
            return {}
        c = {name: last_per_kernel(rd_csv if "RDREQ" in name else wr_csv, name)
             for name in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")}
        out = collections.OrderedDict()
        for k in c["TCC_EA0_RDREQ_sum"]:
            if not k.startswith("vtmc::"):
                continue
            r32, r64, r128 = (c["TCC_EA0_RDREQ_%s_sum" % w].get(k, 0.0) for w in ("32B", "64B", "128B"))
            w_all, w64 = c["TCC_EA0_WRREQ_sum"].get(k, 0.0), c["TCC_EA0_WRREQ_64B_sum"].get(k, 0.0)
            out[k] = {"read_requests": {"all": int(c["TCC_EA0_RDREQ_sum"][k]), "32B": int(r32), "64B": int(r64), "128B": int(r128)},
                      "write_requests": {"all": int(w_all), "64B": int(w64)},
                      "read_bytes": int(32 * r32 + 64 * r64 + 128 * r128), "write_bytes": int(64 * w64 + 32 * (w_all - w64)),
                      "FETCH_SIZE_would_report_bytes": int(64 * c["TCC_EA0_RDREQ_sum"][k])}
        return out

    soup = sized_requests("req_rd", "req_wr")
    json.dump({"kernels": soup, "note": "last dispatch of each kernel in a bench.py run, memory-side (TCC -> fabric) requests by size; read_bytes = 32 n32 + 64 n64 + 128 n128: "
                                        "FETCH_SIZE (= all requests x 64 B) reports half of it when the requests are 128 bytes long -- the emit kernel's 40-byte rows are fetched "
                                        "as whole 128-byte lines (98 % of its requests), which settles round 2's open question"},
              open(os.path.join(dst, "pmc_requests_by_size.json"), "w"), indent=1)
    ist = first(os.path.join(src, "indexed", "*", "*_kernel_stats.csv"))
    if ist:
        shutil.copy(ist, os.path.join(dst, "kernel_stats_indexed.csv"))
        json.dump({"kernels": sized_requests("indexed_rd", "indexed_wr"), "note": "tools/ab_bench.py indexed=1: the <true, true, ...> emit kernel and the <true, ...> classify kernel are the indexed pipeline's"},
                  open(os.path.join(dst, "pmc_requests_by_size_indexed.json"), "w"), indent=1)
    calib = {}
    try:
        rows = [json.loads(ln) for ln in open(os.path.join(src, "calib", "mix2.jsonl")) if ln.startswith("{")]
        best = collections.OrderedDict()
        for r in rows:
            best[r["kernel"]] = max(best.get(r["kernel"], 0.0), r["TBps_med"])
        shutil.copy(os.path.join(src, "calib", "mix2.jsonl"), os.path.join(dst, "mix2_calibration.jsonl"))
        if os.path.exists(os.path.join(src, "calib", "chunks.jsonl")):
            shutil.copy(os.path.join(src, "calib", "chunks.jsonl"), os.path.join(dst, "mix2_chunked_writes.jsonl"))
        calib = {"best_median_TBps_by_mix": best,
                 "note": "tools/calib/mix2: persistent grids of 1-8 workgroups per CU, 1-8 float4 in flight per lane, plain and non-temporal stores; the best median of every read : write mix"}
        # round 4: the plainest kernels there are (one float4 per thread, the grid as large as the buffer) -- they beat every persistent shape
        plain_f = os.path.join(src, "calib", "plain.jsonl")
        if os.path.exists(plain_f):
            shutil.copy(plain_f, os.path.join(dst, "mix2_plain_kernels.jsonl"))
            plain = collections.OrderedDict()
            for r in (json.loads(ln) for ln in open(plain_f) if ln.startswith("{")):
                if r["GiB_total"] >= 2.0:   # beyond the 256 MB Infinity Cache
                    plain[r["kernel"]] = max(plain.get(r["kernel"], 0.0), r["TBps_med"])
            calib["plain_kernels_TBps"] = plain
            calib["note_plain"] = ("tools/calib/mix2 <GiB> copy: one float4 per thread, non-persistent (copy_plain_4 / _8: that many float4 a grid-stride apart), "
                                   "hipMemcpy device to device; best median over buffers of 2 GiB and more")
        fronts_f = os.path.join(src, "calib", "fronts.jsonl")
        if os.path.exists(fronts_f):
            shutil.copy(fronts_f, os.path.join(dst, "mix2_read_fronts.jsonl"))
            calib["read_TBps_by_concurrent_fronts"] = collections.OrderedDict(
                (str(r["fronts"]), r["TBps_med"]) for r in (json.loads(ln) for ln in open(fronts_f) if ln.startswith("{")))
        json.dump(calib, open(os.path.join(dst, "memory_ceilings.json"), "w"), indent=1)
    except Exception as e:   # noqa: BLE001
        print("no calibration:", e)
    for name in ("pytest_gpu.log", "emit_phases.txt", "classify_timeline.txt", "dropin_route.txt"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for name in ("ab_emit_variants.txt", "ab_emit_ablation.txt"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    sys.path.insert(0, ROOT)
    from volumetricterrain_amd import build as vt_build
    traffic = {"kernel_source_sha256": vt_build.kernel_source_hash()}
    for k, v in soup.items():
        name = k.split("::")[1].split("<")[0]
        traffic[name + "_hbm_bytes"] = v["read_bytes"] + v["write_bytes"]
        traffic[name + "_read_bytes"] = v["read_bytes"]
        traffic[name + "_write_bytes"] = v["write_bytes"]
    traffic["source"] = ("profiles/%s/pmc_requests_by_size.json: memory-side requests by size (32 / 64 / 128 B), reads + writes, last dispatch of a bench.py run on the "
                         "builder's lease; valid for the kernel sources with the recorded sha256 only" % tag)
    if calib:
        traffic["memory_ceilings_TBps"] = calib["best_median_TBps_by_mix"]
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "kernel_stats.csv")).read())
    print(json.dumps(traffic, indent=1))
    print(json.dumps(calib, indent=1))
    print("NOTE: profiles/pmc_traffic.json is valid for the CURRENT kernel sources only (sha256 recorded); re-run after any kernel change")


if __name__ == "__main__":
    main()
