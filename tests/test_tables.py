"""Lookup tables: bit-exact against the digests of VoxelTerrain.cs:489-794."""
import hashlib

import numpy as np

DIGEST_EDGE = "ffc58719f11be7a8b34988740a15dcd043dc314a3e2fe01e917fd796001815b9"
DIGEST_TRINUM = "c3ae8bc49cfb9bece3b576fdb62e0da19d1f42c4d719b221bf04c364e02c492a"
DIGEST_VERT = "339ffd018b03993de1e50d0f87171ff286041fb48eaa1f05272b89bbb2275d0e"
EDGE_CONNECTION = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4),
                   (0, 4), (1, 5), (2, 6), (3, 7)]   # MarchingCube.compute:40-43


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, "<i4").tobytes()).hexdigest()


def test_table_digests(oracle_mod):
    edge, tri_num, vert = oracle_mod.tables()
    assert sha(edge) == DIGEST_EDGE
    assert sha(tri_num) == DIGEST_TRINUM
    assert sha(vert) == DIGEST_VERT


def test_table_invariants(oracle_mod):
    edge, tri_num, vert = oracle_mod.tables()
    assert tri_num.sum() == 820 and tri_num.max() == 5
    for c in range(256):
        n = int((vert[c, 0::3] >= 0).sum())
        assert n == tri_num[c]
        used = 0
        for v in vert[c]:
            if v >= 0:
                used |= 1 << int(v)
        geo = 0
        for e, (a, b) in enumerate(EDGE_CONNECTION):
            if ((c >> a) & 1) != ((c >> b) & 1):
                geo |= 1 << e
        assert used == edge[c] == geo
    # every mixed case yields at least one triangle: the streaming classify derives a block's row mask
    # from "case is neither 0 nor 255", the per-block classify from "triangle count != 0" -- the same set
    assert tri_num[0] == 0 and tri_num[255] == 0 and (tri_num[1:255] > 0).all()
