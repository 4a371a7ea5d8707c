"""Device-resident terrain (vtmc_terrain_*): VoxelTerrain.Init's grid and VoxelTerrain.Update's
density write + dirty-block selection (VoxelTerrain.cs:121-149, 262-325) against the CPU
restatement oracle/terrain_ref.c, and the restatement itself against analytic answers.

Grid samples are FP32 results of the reference's own expression order: the bar is bit-exact.
"""
import numpy as np
import pytest

import volumetricterrain_amd as vt


SPECS = [
    ("plane", (9.375, (0, 0), (70, 70), True)),
    ("sphere", ((20.5, 10.25, 30.0), 7.5, True)),
    ("sphere", ((40.0, 9.0, 20.0), 6.0, False)),                       # erode: the mouse edit of SceneManager.cs:121-129
    ("cylinder", ((5.0, 12.0, 5.0), (1.0, 0.25, 0.5), 30.0, 3.0, False)),  # a river bed, RiverRenderer.cs:151-170
    ("sphere", ((-50.0, -50.0, -50.0), 3.0, True)),                    # entirely outside: no samples, no blocks
]


def island_heightmap(res=(48, 40)):
    """A smooth synthetic island: what Island.GetElevation would have filled in (IslandModifier.cs:85-91)."""
    u = np.linspace(-1, 1, res[0], dtype=np.float32)[:, None]
    v = np.linspace(-1, 1, res[1], dtype=np.float32)[None, :]
    return (14.0 * np.exp(-2.5 * (u * u + v * v)) + 1.5 * np.sin(5 * u) * np.cos(4 * v) + 3.0).astype(np.float32)


def build(spec):
    kind, args = spec
    return {"plane": vt.PlaneModifier, "sphere": vt.SphereModifier, "cylinder": vt.CylinderModifier,
            "island": vt.IslandModifier}[kind](*args)


def build_oracle(oracle_mod, spec):
    """The same modifier through the oracle's own (independent) bound formulas."""
    kind, args = spec
    return {"plane": oracle_mod.plane_modifier, "sphere": oracle_mod.sphere_modifier,
            "cylinder": oracle_mod.cylinder_modifier, "island": oracle_mod.heightmap_modifier}[kind](*args)


def queue():
    return [build(s) for s in SPECS]


def oracle_mods(oracle_mod, specs):
    return [build_oracle(oracle_mod, s) for s in specs]


def test_host_bounds_match_the_oracles_restatement(oracle_mod):
    for a, b in zip(queue(), oracle_mods(oracle_mod, SPECS)):
        s = a.to_struct()
        assert list(s.lower) == list(b.lower) and list(s.upper) == list(b.upper)
        assert list(s.p) == list(b.p) and s.kind == b.kind and s.add_or_erode == b.add_or_erode


def test_oracle_fill_and_plane_known_answer(oracle_mod):
    """Init: every sample a void value in [-2,-1).  A plane added at non-integer h gives
    2 * W * H triangles, all vertices at y = h, normals exactly +y: the
    random void values above the plane never win the max against h - y > -1."""
    t = oracle_mod.Terrain(32, 16, 24, seed=3)
    assert t.grid.min() >= -2.0 and t.grid.max() < -1.0
    assert len(np.unique(t.grid)) > t.grid.size // 2          # not one constant: gradients stay non-zero
    dirty = t.update([oracle_mod.plane_modifier(5.375, (0, 0), (40, 40))])
    y = np.arange(18, dtype=np.float32)
    col = t.grid[3, :, 7]
    near = np.abs(np.float32(5.375) - y) < 1
    assert np.array_equal(col[near], (np.float32(5.375) - y)[near])
    assert (col[y < 4] >= 1).all() and (col[y < 4] < 2).all()  # clamped to a full value in [1,2)
    assert (dirty[:, 1] == 0).all() and len(dirty) == 4 * 3    # up.y = ceil(6.375) = 7 < 8: the bottom layer only
    tris, _, _ = oracle_mod.extract_grid(t.grid, dirty)
    assert len(tris) == 2 * 32 * 24
    for f in ("p0", "p1", "p2"):
        assert (tris[f][:, 1] == np.float32(5.375)).all()


def test_oracle_dirty_rule_is_inclusive_on_both_ends(oracle_mod):
    """A sphere whose AABB ends exactly on a block face dirties the face-adjacent block too
    (up >= 8b && low <= 8b + 8, VoxelTerrain.cs:311-313)."""
    t = oracle_mod.Terrain(32, 32, 32)
    dirty = t.update([oracle_mod.sphere_modifier((12.0, 12.0, 12.0), 4.0)])   # samples 8..16
    assert sorted(set(dirty[:, 0])) == [0, 1, 2]
    t2 = oracle_mod.Terrain(32, 32, 32)
    dirty2 = t2.update([oracle_mod.sphere_modifier((12.0, 12.0, 12.0), 3.5)])  # samples 8..16 after floor / ceil
    assert np.array_equal(dirty, dirty2)
    t3 = oracle_mod.Terrain(32, 32, 32)
    dirty3 = t3.update([oracle_mod.sphere_modifier((12.5, 12.5, 12.5), 3.0)])  # samples 9..16: block 0 no longer touched
    assert sorted(set(dirty3[:, 0])) == [1, 2]


def test_oracle_erode_carves_and_reclamps(oracle_mod):
    t = oracle_mod.Terrain(32, 32, 32, seed=5)
    t.update([oracle_mod.plane_modifier(20.5, (0, 0), (40, 40))])
    solid_before = (t.grid > 0).sum()
    t.update([oracle_mod.sphere_modifier((16.0, 16.0, 16.0), 6.0, add=False)])
    assert (t.grid > 0).sum() < solid_before
    assert t.grid[16, 16, 16] < -1.0          # centre: -clamp(6) = -full -> Min -> in (-2,-1]
    assert t.grid.min() >= -2.0 and t.grid.max() < 2.0


@pytest.mark.gpu
def test_gpu_terrain_matches_oracle_bitwise(oracle_mod):
    import torch
    assert torch.cuda.is_available()
    dims, scale, origin, seed = (64, 24, 48), 1.0, (0.0, 0.0, 0.0), 1234
    mods = queue()
    with vt.Extractor(0) as ex:
        ex.terrain_init(*dims, scale, origin, seed)
        ref = oracle_mod.Terrain(*dims, scale, origin, seed)
        assert np.array_equal(ex.terrain_read_samples(), ref.grid)
        # the world build: plane + spheres in one Update
        n_dirty, T = ex.terrain_update(mods[:2])
        want_dirty = ref.update(oracle_mods(oracle_mod, SPECS[:2]))
        assert np.array_equal(ex.terrain_read_samples(), ref.grid)
        assert np.array_equal(ex.terrain_read_samples("z"), ref.grid)   # C# float[,,] layout, same values
        assert n_dirty == len(want_dirty) and np.array_equal(ex.terrain_dirty_blocks(), want_dirty)
        want, want_offs, _ = oracle_mod.extract_grid(ref.grid, want_dirty)
        assert T == len(want) and T > 0
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, want_offs) and np.array_equal(got["block"], want["block"])
        for f in ("p0", "p1", "p2", "n0", "n1", "n2"):
            assert np.abs(got[f] - want[f]).max() <= 1e-5
        # interactive edits: erosions + a modifier outside the world, one Update each
        for spec in SPECS[2:]:
            n_dirty, T = ex.terrain_update([build(spec)])
            want_dirty = ref.update(oracle_mods(oracle_mod, [spec]))
            assert np.array_equal(ex.terrain_read_samples(), ref.grid)
            assert np.array_equal(ex.terrain_dirty_blocks(), want_dirty)
            want, want_offs, _ = oracle_mod.extract_grid(ref.grid, want_dirty)
            assert T == len(want)
            if T:
                got, offs = ex.read_triangles()
                assert np.array_equal(offs, want_offs)
                for f in ("p0", "p1", "p2", "n0", "n1", "n2"):
                    assert np.abs(got[f] - want[f]).max() <= 1e-5
        assert ex.terrain_update([]) == (0, 0)   # empty queue: no BatchUpdate (VoxelTerrain.cs:322)


@pytest.mark.gpu
def test_gpu_terrain_scaled_world_and_full_rebuild(oracle_mod):
    """voxelScale != 1 and a shifted origin; a plane covering the whole world dirties every block,
    which takes the dense streaming path (block ids = the canonical dense order)."""
    dims, scale, origin, seed = (64, 32, 64), 0.5, (-3.0, 1.5, 2.0), 99
    specs = [("plane", (8.3, (-10, -10), (60, 60), True)),
             ("sphere", ((10.0, 40.0, 20.0), 30.0, True))]      # reaches every block together with the plane
    with vt.Extractor(0) as ex:
        ex.terrain_init(*dims, scale, origin, seed)
        ref = oracle_mod.Terrain(*dims, scale, origin, seed)
        n_dirty, T = ex.terrain_update([build(sp) for sp in specs])
        want_dirty = ref.update(oracle_mods(oracle_mod, specs))
        assert n_dirty == len(want_dirty) == 8 * 4 * 8
        assert np.array_equal(ex.terrain_read_samples(), ref.grid)
        want, want_offs, _ = oracle_mod.extract_grid(ref.grid, want_dirty, threads=8)
        assert T == len(want)
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, want_offs) and np.array_equal(got["block"], want["block"])
        for f in ("p0", "p1", "p2", "n0", "n1", "n2"):
            assert np.abs(got[f] - want[f]).max() <= 1e-5


@pytest.mark.gpu
def test_gpu_terrain_errors():
    with vt.Extractor(0) as ex:
        with pytest.raises(vt.VtmcError) as e:
            ex.terrain_update([])
        assert e.value.code == -5
        with pytest.raises(vt.VtmcError) as e:
            ex.terrain_init(20, 16, 16)
        assert e.value.code == -2 and "block size must align" in str(e.value)      # VoxelTerrain.cs:138-139
        with pytest.raises(vt.VtmcError) as e:
            ex.terrain_init(1032, 16, 16)
        assert e.value.code == -2 and "too high resolution" in str(e.value)        # VoxelTerrain.cs:141-142
        # a malformed (inverted, saturating) AABB is an empty range -- the reference's loops would not
        # execute (VoxelTerrain.cs:284-286) -- never a wrapped extent that launches out of bounds
        ex.terrain_init(32, 16, 32, 1.0, (0.0, 0.0, 0.0), 5)
        before = ex.terrain_read_samples().copy()
        m = vt.SphereModifier((10.0, 8.0, 10.0), 4.0, True).to_struct()
        for k in range(3):
            m.lower[k] = 5.0
            m.upper[k] = -3.0e9
        n_dirty, T = ex.terrain_update([m])
        assert n_dirty == 0 and T == 0
        assert np.array_equal(ex.terrain_read_samples(), before)


def test_oracle_heightmap_modifier_known_answer(oracle_mod):
    """A flat heightmap of height h is a plane; a ramp along u is reproduced exactly at the heightmap's
    own sample positions (IslandModifier.cs:45-73)."""
    t = oracle_mod.Terrain(32, 16, 32, seed=2)
    t.update([oracle_mod.heightmap_modifier(np.full((5, 7), 6.25, np.float32), 32.0, 32.0, 10.0)])
    y = np.arange(18, dtype=np.float32)
    near = np.abs(np.float32(6.25) - y) < 1
    assert np.array_equal(t.grid[9, :, 20][near], (np.float32(6.25) - y)[near])
    ramp = np.linspace(2.0, 10.0, 33, dtype=np.float32)[:, None].repeat(4, 1)   # heightmap sample i sits at x = i
    t2 = oracle_mod.Terrain(32, 16, 32, seed=2)
    t2.update([oracle_mod.heightmap_modifier(ramp, 32.0, 32.0, 12.0)])
    for x in (0, 5, 17, 32):
        col = t2.grid[x, :, 11]
        k = int(np.floor(ramp[x, 0]))
        assert col[k] == ramp[x, 0] - np.float32(k)        # unclamped samples next to the surface are exact


@pytest.mark.gpu
def test_gpu_world_build_with_island_heightmap(oracle_mod):
    """The reference's world build (TerrainEngine.cs:87-99): IslandModifier first, then river
    cylinders, one Update.  Grid bit-exact, every block dirty -> dense streaming path."""
    dims, scale, origin, seed = (64, 32, 64), 1.0, (0.0, 0.0, 0.0), 77
    hm = island_heightmap()
    specs = [("island", (hm, 64.0, 64.0, 40.0, True)),
             ("cylinder", ((8.0, 14.0, 10.0), (1.0, -0.1, 0.6), 40.0, 2.5, False)),
             ("cylinder", ((30.0, 12.0, 50.0), (0.3, -0.05, -1.0), 35.0, 2.0, False))]
    with vt.Extractor(0) as ex:
        ex.terrain_init(*dims, scale, origin, seed)
        ref = oracle_mod.Terrain(*dims, scale, origin, seed)
        n_dirty, T = ex.terrain_update([build(s) for s in specs])
        want_dirty = ref.update(oracle_mods(oracle_mod, specs))
        assert np.array_equal(ex.terrain_read_samples(), ref.grid)
        assert n_dirty == len(want_dirty) == 8 * 4 * 8
        want, want_offs, _ = oracle_mod.extract_grid(ref.grid, want_dirty, threads=8)
        assert T == len(want) and T > 5000
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, want_offs) and np.array_equal(got["block"], want["block"])
        for f in ("p0", "p1", "p2", "n0", "n1", "n2"):
            assert np.abs(got[f] - want[f]).max() <= 1e-5
