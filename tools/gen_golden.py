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
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
