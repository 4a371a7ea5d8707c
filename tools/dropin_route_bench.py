#!/usr/bin/env python3
"""The literal drop-in route, measured: what BatchUpdate (VoxelTerrain.cs:330-477) becomes when a C# host keeps its float[W+2, E+2, H+2]
grid and calls vtmc_extract_grid with its dirty list + vtmc_read_triangles.  World = the reference's demo scene (256 x 72 x 256 cells,
SceneManager.cs:23-24, voxelScale 1), z fastest as a C# float[,,]; density = a heightfield-like plane + sphere edits on the CPU oracle's
terrain twin.  Reported per case: wall time of extract_grid (host gather or upload + kernels + the one wait), of read_triangles (PCIe
down), and the device-side stage times.

    python tools/dropin_route_bench.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import oracle
    import volumetricterrain_amd as vt
    W, E, H = 256, 72, 256
    t = oracle.Terrain(W, E, H, seed=3)
    t.update([oracle.plane_modifier(30.5, (0.0, 0.0), (float(W), float(H))),
              oracle.sphere_modifier((128.0, 40.0, 128.0), 24.0), oracle.sphere_modifier((60.0, 30.0, 200.0), 14.0, add=False)])
    grid = np.ascontiguousarray(t.grid)            # [x, y, z], z fastest: the C# float[,,] order
    nb = (W // 8, E // 8, H // 8)
    all_blocks = oracle.all_blocks(W, E, H)
    ex = vt.Extractor(0)

    def case(name, blocks, reps=20):
        for _ in range(3):
            ex.extract_grid(grid, blocks)
            ex.read_triangles()
        te, tr, T = [], [], 0
        for _ in range(reps):
            t0 = time.perf_counter()
            T = ex.extract_grid(grid, blocks)
            t1 = time.perf_counter()
            ex.read_triangles()
            t2 = time.perf_counter()
            te.append((t1 - t0) * 1e3)
            tr.append((t2 - t1) * 1e3)
        ms = ex.last_stage_ms()
        n = len(blocks) if blocks is not None else nb[0] * nb[1] * nb[2]
        print("%-44s %6d blocks  T=%8d  extract_grid %8.3f ms (device %6.3f)  read_triangles %7.3f ms"
              % (name, n, T, float(np.median(te)), ms["total"], float(np.median(tr))), flush=True)

    case("whole world, no list (world build)", None)
    case("whole world as a dirty list", all_blocks)
    # an interactive edit: sphere r = 10 (SceneManager.cs:121-129) -> the blocks its AABB touches
    lo, hi = np.array([118, 30, 118]) // 8, np.array([138, 50, 138]) // 8 + 1
    edit = np.array([[x, y, z] for z in range(lo[2], hi[2]) for y in range(lo[1], hi[1]) for x in range(lo[0], hi[0])], np.int32)
    case("edit: sphere r = 10 (dirty list)", edit)
    mid = all_blocks[(all_blocks[:, 1] >= 2) & (all_blocks[:, 1] <= 4)][::3]
    case("a third of the surface band (dirty list)", np.ascontiguousarray(mid))
    ex.close()


if __name__ == "__main__":
    main()
