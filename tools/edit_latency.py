#!/usr/bin/env python3
"""Latency of the reference's interactive edit (SceneManager.cs:121-129 -> TerrainEngine.ModifyTerrain
-> InsertModifier(SphereModifier r=10) -> next frame's VoxelTerrain.Update) on the default demo world
(256 x 72 x 256 cells, SceneManager.cs:23-24) with the grid resident in HBM: one vtmc_terrain_update
per edit = density write + dirty set + classify + scan + emit + read-back of T."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import volumetricterrain_amd as vt


def main():
    rng = np.random.default_rng(1)
    with vt.Extractor(0) as ex:
        ex.terrain_init(256, 72, 256, 1.0, (0.0, 0.0, 0.0), 1)
        t0 = time.perf_counter()
        nd, T = ex.terrain_update([vt.PlaneModifier(30.5, (-1, -1), (300, 300), True)])
        build_s = time.perf_counter() - t0
        lat, tris, blocks = [], [], []
        for i in range(220):
            c = (float(rng.uniform(20, 236)), 30.0 + float(rng.uniform(-4, 4)), float(rng.uniform(20, 236)))
            m = vt.SphereModifier(c, 10.0, bool(i & 1))
            t0 = time.perf_counter()
            nd, T = ex.terrain_update([m])
            dt = time.perf_counter() - t0
            if i >= 20:
                lat.append(dt)
                tris.append(T)
                blocks.append(nd)
                stage = ex.last_stage_ms()
        lat = np.array(lat) * 1e6
        print(json.dumps({"world": "256x72x256 cells, plane + 200 sphere edits r=10 (alternating add / erode)",
                          "world_build_ms": round(build_s * 1e3, 3), "edit_latency_us_median": round(float(np.median(lat)), 1),
                          "edit_latency_us_p90": round(float(np.percentile(lat, 90)), 1),
                          "dirty_blocks_median": int(np.median(blocks)), "triangles_median": int(np.median(tris)),
                          "last_stage_ms": stage}))


if __name__ == "__main__":
    main()
