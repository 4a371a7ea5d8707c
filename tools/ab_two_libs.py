#!/usr/bin/env python3
"""Same-box, same-PROCESS A/B of library builds: boxes of the pool drift by several percent from one process to the next (clock / power
state), more than most kernel changes are worth, so alternating processes cannot resolve them.  Round 6: the build listed FIRST draws the first context's
output buffers, whose allocation decides the emit kernel's level (profiles/r06/placement_probe.txt) -- run both orders, or give every context
placement trials (append place_outputs=8 to a variant).  Every build is
loaded beside the others (its own ctypes handle, its own context), the rounds alternate build by build and variant by variant.

    python tools/ab_two_libs.py new=volumetricterrain_amd/libvtmc.so r05=tools/_ab/libvtmc_r05.so -- base indexed=1 emit_once=0 [--rounds 9]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    argv = sys.argv[1:]
    cut = argv.index("--") if "--" in argv else len(argv)
    libs = [a.split("=", 1) for a in argv[:cut]]
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*", default=["base"])
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--limit", type=int, default=0)
    args = ap.parse_args(argv[cut + 1:])
    import torch
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding

    n, c = args.n, args.chunk
    dim = c + 2
    origins = sharding.chunk_origins(n, c)
    if args.limit:
        origins = origins[:args.limit]
    exs = {name: vt.Extractor(0, lib_path=os.path.join(ROOT, path) if not os.path.isabs(path) else path) for name, path in libs}
    first = next(iter(exs.values()))
    d = torch.empty(len(origins) * dim ** 3, dtype=torch.float32, device="cuda")
    first.density_fill_device(vt.density_params("perlin3d", n), origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())

    def apply(ex, spec):
        kv = {} if spec == "base" else dict((k, int(v)) for k, v in (it.split("=") for it in spec.split(",")))
        ex.set_output_mode(bool(kv.pop("indexed", 0)))
        return kv

    def run(ex, spec):
        kv = apply(ex, spec)
        ex.set_tuning(**kv)
        T = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(origins), dim ** 3, None, 0)
        ms = ex.last_stage_ms()
        for k in kv:   # back to the build's own defaults: emit_once / fast math etc. are 1 / 1, everything else set here is restored by hand
            ex.set_tuning(**{k: {"emit_once": 1, "emit_fast_math": 1, "emit_dynamic": 1, "emit_row_masks": 1, "emit_sub_log2": 1, "classify_wgs_per_cu": 3}.get(k, 0)})
        return T, ms

    for ex in exs.values():
        for v in args.variants:
            run(ex, v)
            run(ex, v)
    res = {(ln, v): {"classify": [], "scan": [], "emit": [], "total": []} for ln in exs for v in args.variants}
    Ts = {}
    for _ in range(args.rounds):
        for v in args.variants:
            for ln, ex in exs.items():
                T, ms = run(ex, v)
                Ts[(ln, v)] = T
                for k in res[(ln, v)]:
                    res[(ln, v)][k].append(ms[k])
    for v in args.variants:
        for ln in exs:
            r = res[(ln, v)]
            print("%-10s %-28s T %d  " % (ln, v, Ts[(ln, v)]) + "  ".join("%s med %.4f min %.4f" % (k, statistics.median(x), min(x)) for k, x in r.items()))


if __name__ == "__main__":
    main()
