#!/usr/bin/env python3
"""Does the emit kernel's time depend on WHERE its output buffer lies?  (round 6: in same-process A/Bs of library builds the context created
first ran the IDENTICAL soup kernel 13 % slower than the others -- 1.00 against 0.88 ms, the very spread rounds 4-5 put down to "the boxes
differ".)  Several contexts of ONE library in one process, the same resident field, alternating rounds; per context: the device address of its
triangle buffer and its stage times; then context 0's buffer is released and re-reserved a few times (a new address each time) and timed again.
    python tools/placement_probe.py [--contexts 6] [--rounds 9]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--contexts", type=int, default=6)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--field-first", action="store_true", help="allocate the density field before the contexts exist")
ap.add_argument("--tune", default="", help="key=value,... for every context")
ap.add_argument("--no-realloc", action="store_true")
ap.add_argument("--lib", default=None, help="another build of the library (path)")
ap.add_argument("--fields", type=int, default=0, help="also: this many COPIES of the input field at other addresses, every context timed on every copy")
a = ap.parse_args()
n, c, dim = 1024, 128, 130
org = sharding.chunk_origins(n, c)


def field():
    return torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")


d = field() if a.field_first else None
exs = [vt.Extractor(0, lib_path=a.lib) for _ in range(a.contexts)]
if a.tune:
    for e in exs:
        e.set_tuning(**{k: int(v) for k, v in (it.split("=") for it in a.tune.split(","))})
if d is None:
    d = field()
exs[0].density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())


def step(e):
    T = e.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3)
    return T, e.last_stage_ms()


def measure(ctxs, rounds):
    res = {id(e): {"classify": [], "emit": []} for e in ctxs}
    for _ in range(rounds):
        for e in ctxs:
            _, ms = step(e)
            res[id(e)]["classify"].append(ms["classify"])
            res[id(e)]["emit"].append(ms["emit"])
    return res


for e in exs:
    step(e)
    step(e)
print("field at 0x%x (%d MB)   %d contexts, %d rounds alternating" % (d.data_ptr(), d.numel() * 4 >> 20, len(exs), a.rounds))
res = measure(exs, a.rounds)
for i, e in enumerate(exs):
    tri, off, _ = e.device_results()
    r = res[id(e)]
    print("context %d: triangles at 0x%012x  (mod 2 MiB %7d, mod 1 GiB %4d MiB)   offsets at 0x%012x   classify med %.4f   emit med %.4f min %.4f"
          % (i, tri, tri % (2 << 20), (tri % (1 << 30)) >> 20, off, statistics.median(r["classify"]), statistics.median(r["emit"]), min(r["emit"])))
if a.fields:
    # is a level a property of the output buffer alone, or of the pair (input field, output buffer)?
    copies = [d] + [d.clone() for _ in range(a.fields)]
    print("every context on %d copies of the field (rows: contexts, columns: the field at %s); emit med ms" % (len(copies), " ".join("0x%x" % c_.data_ptr() for c_ in copies)))
    table = {(i, k): [] for i in range(len(exs)) for k in range(len(copies))}
    ctab = {(i, k): [] for i in range(len(exs)) for k in range(len(copies))}
    for _ in range(a.rounds):
        for k, dk in enumerate(copies):
            for i, e in enumerate(exs):
                e.extract_volumes_device(dk.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3)
                table[(i, k)].append(e.last_stage_ms()["emit"])
                ctab[(i, k)].append(e.last_stage_ms()["classify"])
    for i in range(len(exs)):
        print("  context %d: %s" % (i, "  ".join("%.4f" % statistics.median(table[(i, k)]) for k in range(len(copies)))))
    print("the classify kernel on the same copies (one read stream over the field: does IT care where the field lies?)")
    for i in range(len(exs)):
        print("  context %d: %s" % (i, "  ".join("%.4f" % statistics.median(ctab[(i, k)]) for k in range(len(copies)))))
    del copies
if a.no_realloc:
    for e in exs:
        e.close()
    vt.release_streams()   # profiled (tools/placement_counters.py): no stream of the library's alive when the profiler finalises
    sys.exit(0)
print("context 0's triangle buffer released and re-reserved (vtmc_reserve_triangles), timed beside context 1:")
T, _ = step(exs[0])
for k, cap in enumerate([T + 1000, T + T // 8 + 1024, T + T // 4, T + 4096, 2 * T, T + T // 8 + 1024]):
    exs[0].reserve_triangles(cap)
    step(exs[0])
    res = measure(exs[:2], a.rounds)
    tri, _, _ = exs[0].device_results()
    print("  capacity %9d: triangles at 0x%012x (mod 2 MiB %7d)   emit med %.4f min %.4f    | context 1 beside it: emit med %.4f"
          % (cap, tri, tri % (2 << 20), statistics.median(res[id(exs[0])]["emit"]), min(res[id(exs[0])]["emit"]), statistics.median(res[id(exs[1])]["emit"])))
