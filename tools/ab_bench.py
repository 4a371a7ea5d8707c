#!/usr/bin/env python3
"""A/B harness: interleaved rounds of kernel variants in ONE process on the bench workload
(cdna_hip_programming.md rule 24).  Prints median / min per stage for each variant.

    python tools/ab_bench.py "emit_assign=0" "emit_assign=1" "emit_assign=1,emit_wide_store=0" --rounds 7
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help="comma-separated key=value tuning sets; 'base' = defaults")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--kind", default="perlin3d")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--limit", type=int, default=0, help="use only the first K chunks")
    args = ap.parse_args()
    import torch
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding

    n, c = args.n, args.chunk
    dim = c + 2
    origins = sharding.chunk_origins(n, c)
    if args.limit:
        origins = origins[:args.limit]
    ex = vt.Extractor(0)
    d = torch.empty(len(origins) * dim ** 3, dtype=torch.float32, device="cuda")
    ex.density_fill_device(vt.density_params(args.kind, n), origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3,
                           d.data_ptr())
    defaults = dict(emit_fast_math=1, emit_wgs_per_cu=0, emit_dynamic=1, emit_sub_log2=1, indexed=0, emit_row_masks=1, classify_wgs_per_cu=3, stage_events=1, emit_once=1)
    if os.environ.get("VTMC_LIB", "").endswith(("_diag.so", "_phases.so", "_timeline.so")):   # -DVTMC_DIAGNOSTICS builds know the ablation keys
        defaults.update(emit_ablate=0, classify_ablate=0)

    def apply(spec):
        kv = dict(defaults)
        if spec != "base":
            for item in spec.split(","):
                k, v = item.split("=")
                kv[k] = int(v)
        ex.set_output_mode(bool(kv.pop("indexed", 0)))
        ex.set_tuning(**kv)

    def run():
        T = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(origins), dim ** 3,
                                      None, args.flags)
        return T, ex.last_stage_ms()

    apply("base")
    for _ in range(2):
        run()
    res = {v: {"classify": [], "scan": [], "emit": [], "total": []} for v in args.variants}
    for _ in range(args.rounds):
        for v in args.variants:
            apply(v)
            T, ms = run()
            for k in res[v]:
                res[v][k].append(ms[k])
    print("T =", T)
    for v in args.variants:
        print("%-48s" % v, "  ".join("%s med %.4f min %.4f" % (k, statistics.median(x), min(x)) for k, x in res[v].items()))
    ex.close()
    vt.release_streams()   # this tool runs under rocprofv3: no stream of the library's may be alive when the profiler finalises


if __name__ == "__main__":
    main()
