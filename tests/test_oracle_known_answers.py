"""Pins the CPU oracle with the analytic answers of SURVEY.md section 4 (the reference holds no
tests or golden vectors of its own): plane, sphere, empty / full, block-decomposition invariance,
layout independence, and agreement of the staged (three-dispatch) and fused oracle paths."""
import numpy as np
import pytest

import fields


def tri_key(tris):
    """Canonical sortable view of a triangle multiset (per block)."""
    a = np.zeros((len(tris), 19), np.float64)
    a[:, 0] = tris["block"]
    for i, f in enumerate(("p0", "p1", "p2", "n0", "n1", "n2")):
        a[:, 1 + 3 * i:4 + 3 * i] = tris[f]
    return a[np.lexsort(a.T[::-1])]


def test_plane_known_answer(oracle_mod):
    n = (16, 16, 24)
    h = 5.375
    g = fields.plane(n, h)
    tris, offs, cases = oracle_mod.extract_grid(g, want_cases=True)
    # every cell of the layer holding h is case 0x33 => 2 triangles (table row 51)
    assert len(tris) == 2 * n[0] * n[2]
    blocks = oracle_mod.all_blocks(*n)
    layer = cases.reshape(len(blocks), 8, 8, 8)   # [b, z, y, x]
    for b, (bx, by, bz) in enumerate(blocks):
        for y in range(8):
            gy = by * 8 + y
            want = 0x33 if gy == 5 else (0xFF if gy < 5 else 0x00)
            assert (layer[b, :, y, :] == want).all()
    for f in ("p0", "p1", "p2"):
        assert np.allclose(tris[f][:, 1], 5.375 % 8, atol=0, rtol=0)
    for f in ("n0", "n1", "n2"):
        assert (tris[f] == np.array([0, 1, 0], np.float32)).all()
    # winding agrees with the normal: cross(p1-p0, p2-p0) points +y
    cr = np.cross(tris["p1"] - tris["p0"], tris["p2"] - tris["p0"])
    assert (cr[:, 1] > 0).all() and np.allclose(cr[:, [0, 2]], 0)


def test_sphere_known_answer(oracle_mod):
    n = (32, 32, 32)
    c, r = (16.3, 15.6, 16.9), 10.2
    tris, offs, _ = oracle_mod.extract_grid(fields.sphere(n, c, r))
    assert len(tris) > 1000
    blocks = oracle_mod.all_blocks(*n)
    org = blocks[tris["block"]].astype(np.float64) * 8
    for f, nf in (("p0", "n0"), ("p1", "n1"), ("p2", "n2")):
        p = tris[f].astype(np.float64) + org
        d = np.linalg.norm(p - np.array(c), axis=1)
        assert np.abs(d - r).max() < 0.06          # linear-interpolation error of a curved field
        radial = (p - np.array(c)) / d[:, None]
        nn = tris[nf].astype(np.float64)
        cosang = (nn * radial).sum(1) / np.linalg.norm(nn, axis=1)
        assert cosang.min() > 0.97                  # outward, forward-difference accuracy
    # closed surface: welded mesh has Euler characteristic 2
    allp = np.concatenate([tris[f].astype(np.float64) + org for f in ("p0", "p1", "p2")])
    key = np.round(allp * 4096).astype(np.int64)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    T = len(tris)
    tri_idx = inv.reshape(3, T).T
    e = np.concatenate([tri_idx[:, [0, 1]], tri_idx[:, [1, 2]], tri_idx[:, [2, 0]]])
    e.sort(axis=1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all()
    assert len(uniq) - len(ue) + T == 2


@pytest.mark.parametrize("value", [-1.5, 0.0, 1.5, np.nan])
def test_empty_and_full(oracle_mod, value):
    tris, offs, cases = oracle_mod.extract_grid(fields.constant((8, 16, 8), value), want_cases=True)
    assert len(tris) == 0 and (offs == 0).all()
    assert (cases == (0xFF if value > 0 else 0)).all()   # strict '>' ; NaN => outside


def test_staged_equals_fused(oracle_mod):
    g = oracle_mod.density_volume("perlin3d", 32)
    blocks = oracle_mod.all_blocks(32, 32, 32)
    tiles = oracle_mod.gather_tiles(g, blocks)
    t1, o1, c1 = oracle_mod.extract_tiles(tiles)
    t2, o2, c2 = oracle_mod.extract_grid(g, want_cases=True)
    assert (o1 == o2).all() and (c1 == c2).all() and t1.tobytes() == t2.tobytes()
    t3, o3, _ = oracle_mod.extract_grid(g, threads=4)
    assert t3.tobytes() == t2.tobytes() and (o3 == o2).all()


def test_block_decomposition_invariance(oracle_mod):
    """the triangle multiset does not depend on how the grid is batched."""
    g = oracle_mod.density_volume("perlin3d", 32)
    blocks = oracle_mod.all_blocks(32, 32, 32)
    whole, _, _ = oracle_mod.extract_grid(g, blocks)
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(blocks))
    parts = []
    for chunk in np.array_split(perm, 5):
        t, _, _ = oracle_mod.extract_grid(g, blocks[chunk])
        t = t.copy()
        t["block"] = chunk[t["block"]]
        parts.append(t)
    merged = np.concatenate(parts)
    assert np.array_equal(tri_key(whole), tri_key(merged), equal_nan=True)


def test_layout_independence(oracle_mod):
    """A C# float[,,] (z fastest, VoxelTerrain.cs:145) and the x-fastest layout give identical output."""
    gx = oracle_mod.density_volume("perlin3d", 16, order="x")
    gz = oracle_mod.density_volume("perlin3d", 16, order="z")
    assert np.array_equal(gx, gz)
    a, _, _ = oracle_mod.extract_grid(gx)
    b, _, _ = oracle_mod.extract_grid(gz)
    assert a.tobytes() == b.tobytes()


def test_all_256_cases_and_bin(oracle_mod):
    tiles = fields.all_cases_tile()
    tris, offs, cases = oracle_mod.extract_tiles(tiles)
    _, tri_num, _ = oracle_mod.tables()
    got = cases.reshape(4, 8, 8, 8)[:, 0::2, 0::2, 0::2].reshape(4, 64)
    assert (got == np.arange(256).reshape(4, 64)).all()
    # isolated cells reproduce exactly their own table count (+ the neighbours they induce)
    assert len(tris) == offs[-1] and len(tris) >= 820
    # binning (VoxelTerrain.cs:437-446): shuffled input, scaled positions, per-block grouping
    rng = np.random.default_rng(0)
    shuffled = tris[rng.permutation(len(tris))]
    verts, nrms, boffs = oracle_mod.bin_triangles(shuffled, 4, voxel_scale=2.0)
    assert (boffs == offs).all()
    for b in range(4):
        sel = shuffled[shuffled["block"] == b]
        assert np.array_equal(verts[boffs[b]:boffs[b + 1], 0], sel["p0"] * np.float32(2.0))
        assert np.array_equal(nrms[boffs[b]:boffs[b + 1], 2], sel["n2"], equal_nan=True)


def test_density_definition(oracle_mod):
    """perlin3d: |noise| <= ~1.04, zero at noise-lattice points, chunked fill equals whole fill."""
    n = 32
    g = oracle_mod.density_volume("perlin3d", n)
    assert np.abs(g).max() < 1.1
    assert g[0, 0, 0] == 0 and g[4, 8, 12] == 0       # f = 8/32: lattice every 4 samples
    sub = oracle_mod.density_volume("perlin3d", n, origin=(8, 16, 0), dims=(10, 10, 10))
    assert np.array_equal(sub, g[8:18, 16:26, 0:10])
    perm = oracle_mod.permutation(1337)
    assert sorted(perm.tolist()) == list(range(256))
    f = oracle_mod.density_volume("fbm8", n)
    assert f[:, 0, :].mean() > 0.5 and f[:, n + 1, :].mean() < -0.5   # solid below, air above
