#!/usr/bin/env python3
"""World build as TerrainEngine.Init does it (TerrainEngine.cs:87-99): one IslandModifier (bilinear
heightmap) plus one CylinderModifier per river segment, then ONE VoxelTerrain.Update -- the
reference's largest serial loop (VoxelTerrain.cs:284-305, one virtual QueryDensity call + two
Random.Range per sample).  Here: vtmc_terrain_update on a grid resident in HBM; beside it the CPU
restatement (oracle/terrain_ref.c, one thread, as the reference's loop is) on a bounded world.
Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import volumetricterrain_amd as vt


def heightmap(res):
    u = np.linspace(-1, 1, res, dtype=np.float32)[:, None]
    v = np.linspace(-1, 1, res, dtype=np.float32)[None, :]
    return (0.55 * np.exp(-2.5 * (u * u + v * v)) + 0.06 * np.sin(7 * u) * np.cos(5 * v) + 0.12).astype(np.float32)


def rivers(rng, n, w, e, h):
    out = []
    for _ in range(n):
        start = (float(rng.uniform(0.2, 0.8) * w), float(rng.uniform(0.25, 0.5) * e), float(rng.uniform(0.2, 0.8) * h))
        d = (float(rng.normal()), float(rng.normal() * 0.1), float(rng.normal()))
        out.append((start, d, float(rng.uniform(0.05, 0.15) * w), float(rng.uniform(1.5, 3.0))))
    return out


def main():
    import oracle
    W, E, H = 1024, 256, 1024
    rng = np.random.default_rng(3)
    hm = heightmap(512)
    riv = rivers(rng, 40, W, E, H)
    mods = [vt.IslandModifier(hm * E, float(W), float(H), float(E), True)] + \
           [vt.CylinderModifier(s, d, L, r, False) for s, d, L, r in riv]
    samples = (W + 2) * (E + 2) * (H + 2)
    owners = mods                            # the structs borrow the heightmap array: keep the objects alive
    mods = [m.to_struct() for m in owners]   # the C# shim fills these structs; not part of the timed call
    with vt.Extractor(0) as ex:
        times = []
        for _ in range(4):
            ex.terrain_init(W, E, H, 1.0, (0.0, 0.0, 0.0), 5)
            t0 = time.perf_counter()
            nd, T = ex.terrain_update(mods)
            times.append(time.perf_counter() - t0)
            stage = ex.last_stage_ms()
        gpu_s = min(times[1:])
    # CPU restatement on a 256 x 64 x 256 world of the same shape (bounded: ~4.4 M samples x 41 modifiers)
    w, e, h = 256, 64, 256
    ref = oracle.Terrain(w, e, h, 1.0, (0.0, 0.0, 0.0), 5)
    rmods = [oracle.heightmap_modifier(hm * e, float(w), float(h), float(e))] + \
            [oracle.cylinder_modifier(tuple(np.array(s) * (w / W)), d, L * (w / W), r, add=False) for s, d, L, r in riv]
    t0 = time.perf_counter()
    ref.update(rmods)
    cpu_s = time.perf_counter() - t0
    cpu_samples = (w + 2) * (e + 2) * (h + 2)
    print(json.dumps({
        "world": "%dx%dx%d cells, IslandModifier (512^2 heightmap) + %d river cylinders, one Update" % (W, E, H, len(riv)),
        "gpu_update_ms": round(gpu_s * 1e3, 3), "gpu_msamples_per_s_whole_update": round(samples / gpu_s / 1e6, 1),
        "extract_stage_ms": stage, "dirty_blocks": nd, "triangles": T,
        "cpu_port": {"world": "%dx%dx%d" % (w, e, h), "seconds": round(cpu_s, 3), "threads": 1,
                     "msamples_per_s": round(cpu_samples / cpu_s / 1e6, 2)}}))


if __name__ == "__main__":
    main()
