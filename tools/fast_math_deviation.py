#!/usr/bin/env python3
"""Largest deviation of the shipped (fast-math) emit kernels from the oracle on BASELINE config[1] (256^3 perlin3d): the vertex-once soup
(default), the per-corner soup and the de-indexed welded output.  Bar: 1e-5 absolute (BASELINE.json north_star)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402

g = oracle.density_volume("perlin3d", 256)
want, offs, _ = oracle.extract_grid(g, threads=oracle.max_threads())
F = ("p0", "p1", "p2", "n0", "n1", "n2")
with vt.Extractor(0) as ex:
    for once in (1, 0):
        ex.set_tuning(emit_once=once)
        assert ex.extract_grid(g) == len(want)
        got, _ = ex.read_triangles()
        print("soup, emit_once=%d: positions %.3g  normals %.3g" % (once, max(float(np.nanmax(np.abs(got[f] - want[f]))) for f in F[:3]),
                                                                   max(float(np.nanmax(np.abs(got[f] - want[f]))) for f in F[3:])))
    ex.set_output_mode(True)
    assert ex.extract_grid(g) == len(want)
    v, i, vo, to = ex.read_indexed_mesh()
    back = oracle.deindex(v.view(oracle.VERTEX_DTYPE), i, vo, to)
    print("indexed, de-indexed: positions %.3g  normals %.3g" % (max(float(np.nanmax(np.abs(back[f] - want[f]))) for f in F[:3]),
                                                                max(float(np.nanmax(np.abs(back[f] - want[f]))) for f in F[3:])))
