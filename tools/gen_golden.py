#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/mc_oracle.c).

The reference holds no golden vectors and cannot be executed here (C# + Unity HLSL),
so these fixtures pin the build's own restatement: inputs + expected cases / offsets / triangles in
the canonical order (block, cell x+8y+64z, table triangle i).  Run:  python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import fields  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    # 1. one perlin3d grid of 16^3 cells (8 blocks), N = 16 => f = 0.5
    g = np.ascontiguousarray(oracle.density_volume("perlin3d", 16).transpose(2, 1, 0))   # memory order z,y,x
    tris, offs, cases = oracle.extract_grid(g.transpose(2, 1, 0), want_cases=True)
    np.savez_compressed(os.path.join(OUT, "perlin16.npz"), grid_zyx=g, triangles=tris, block_tri_offsets=offs,
                        cases=cases)
    # 2. the tile batch that exercises all 256 cube cases
    tiles = fields.all_cases_tile()
    tris, offs, cases = oracle.extract_tiles(tiles)
    np.savez_compressed(os.path.join(OUT, "all_cases_tiles.npz"), tiles=tiles, triangles=tris,
                        block_tri_offsets=offs, cases=cases)
    # 3. sphere carved out of a plane slab: the reference's own modifiers (TerrainModifier.cs:59-62, :79-82)
    n = (16, 16, 16)
    g = np.minimum(fields.plane(n, 9.25), -fields.sphere(n, (8.3, 9.0, 7.6), 4.4))
    g = np.ascontiguousarray(g.transpose(2, 1, 0))
    tris, offs, cases = oracle.extract_grid(g.transpose(2, 1, 0), want_cases=True)
    np.savez_compressed(os.path.join(OUT, "plane_minus_sphere16.npz"), grid_zyx=g, triangles=tris,
                        block_tri_offsets=offs, cases=cases)
    # 4. terrain: Init's fill + one Update of the reference's modifier kinds (plane, sphere add / erode,
    #    cylinder erode, heightmap) on a 32 x 16 x 24 world, scale 0.5, shifted origin; grid, dirty list and
    #    the soup + welded meshes of the dirty blocks
    t = oracle.Terrain(32, 16, 24, 0.5, (-1.0, 0.5, 2.0), seed=20151)
    hm = (3.0 + 2.0 * np.sin(np.arange(7, dtype=np.float32)[:, None] * 0.9) * np.cos(np.arange(5, dtype=np.float32)[None, :] * 1.3)).astype(np.float32)
    spec = np.array([
        # kind, add, then the constructor arguments padded to 9 floats
        (0, 1, 3.3, -5, -5, 40, 40, 0, 0, 0, 0),                # plane(height, low.x, low.y, up.x, up.y)
        (1, 1, 6.0, 4.0, 8.0, 2.6, 0, 0, 0, 0, 0),              # sphere(center, radius)
        (1, 0, 10.0, 3.5, 6.0, 1.9, 0, 0, 0, 0, 0),
        (2, 0, 1.0, 2.5, 3.0, 1.0, 0.2, 0.6, 9.0, 1.1, 0),      # cylinder(start, dir, length, radius)
        (3, 1, 12.0, 10.0, 6.0, 0, 0, 0, 0, 0, 0),              # heightmap(island w, h, max elevation)
    ], np.float32)
    mods = []
    for row in spec:
        k, add, a = int(row[0]), bool(row[1]), row[2:]
        if k == 0:
            mods.append(oracle.plane_modifier(a[0], (a[1], a[2]), (a[3], a[4]), add))
        elif k == 1:
            mods.append(oracle.sphere_modifier(a[0:3], a[3], add))
        elif k == 2:
            mods.append(oracle.cylinder_modifier(a[0:3], a[3:6], a[6], a[7], add))
        else:
            mods.append(oracle.heightmap_modifier(hm, a[0], a[1], a[2], add))
    fill = t.grid.copy()
    dirty = t.update(mods)
    tris, offs, _ = oracle.extract_grid(t.grid, dirty)
    verts, idx, voffs, _ = oracle.extract_grid_indexed(t.grid, dirty)
    np.savez_compressed(os.path.join(OUT, "terrain_update.npz"), modifiers=spec, heightmap=hm,
                        fill_zyx=np.ascontiguousarray(fill.transpose(2, 1, 0)),
                        grid_zyx=np.ascontiguousarray(t.grid.transpose(2, 1, 0)), dirty=dirty, triangles=tris,
                        block_tri_offsets=offs, vertices=verts, indices=idx, block_vertex_offsets=voffs)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
