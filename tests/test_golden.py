"""Committed golden vectors (tests/golden/, made by tools/gen_golden.py with the CPU oracle):
the oracle must keep reproducing them bit for bit (CPU), and the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRIDS = ["perlin16.npz", "plane_minus_sphere16.npz"]


def load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("name", GRIDS)
def test_oracle_reproduces_golden_grid(oracle_mod, name):
    z = load(name)
    tris, offs, cases = oracle_mod.extract_grid(z["grid_zyx"].transpose(2, 1, 0), want_cases=True)
    assert np.array_equal(offs, z["block_tri_offsets"]) and np.array_equal(cases, z["cases"])
    assert tris.tobytes() == z["triangles"].tobytes()


def test_oracle_reproduces_golden_tiles(oracle_mod):
    z = load("all_cases_tiles.npz")
    tris, offs, cases = oracle_mod.extract_tiles(z["tiles"])
    assert np.array_equal(offs, z["block_tri_offsets"]) and np.array_equal(cases, z["cases"])
    assert tris.tobytes() == z["triangles"].tobytes()
    assert sorted(set(z["cases"].ravel().tolist())) == list(range(256))   # every cube case is present


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [False, True])
def test_hip_path_matches_golden(exact):
    import volumetricterrain_amd as vt
    with vt.Extractor(0) as ex:
        ex.set_tuning(emit_fast_math=0 if exact else 1)
        for name in GRIDS + ["all_cases_tiles.npz"]:
            z = load(name)
            if "tiles" in z:
                T = ex.extract_blocks(z["tiles"])
            else:
                T = ex.extract_grid(z["grid_zyx"].transpose(2, 1, 0))
            want = z["triangles"]
            assert T == len(want)
            got, offs = ex.read_triangles()
            assert np.array_equal(offs, z["block_tri_offsets"])
            assert np.array_equal(ex.read_cases(), z["cases"])
            assert np.array_equal(got["block"], want["block"])
            for f in ("p0", "p1", "p2", "n0", "n1", "n2"):
                nan = np.isnan(want[f])
                assert np.array_equal(np.isnan(got[f]), nan)
                dev = np.abs(np.where(nan, 0, got[f]) - np.where(nan, 0, want[f])).max()
                assert dev <= (0.0 if exact else 1e-5), (name, f, dev)
