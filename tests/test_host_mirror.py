"""C++ host mirror of the reference's VoxelTerrain chunk API (host/voxel_terrain.*).
CPU: Init validation, CSG density writes, dirty-block selection (recording backend).
GPU: Init -> InsertModifier x3 -> Update through libvtmc.so, meshes compared with the oracle run on
the very grid the host built."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "_build", "host_selftest")


def build_host():
    import volumetricterrain_amd as vt
    vt.load()   # libvtmc.so must exist to link against
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "host")], check=True)
    return EXE


def test_host_logic_without_gpu():
    exe = build_host()
    r = subprocess.run([exe, "--cpu"], capture_output=True, text=True)
    assert r.returncode == 0 and "HOST-CPU-OK" in r.stdout, r.stderr


@pytest.mark.gpu
def test_host_mirror_end_to_end(tmp_path, oracle_mod):
    exe = build_host()
    r = subprocess.run([exe, "--gpu", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "HOST-GPU-OK" in r.stdout, r.stdout + r.stderr
    W, E, H, scale = 64, 32, 64, np.float32(0.5)
    grid = np.fromfile(tmp_path / "grid.f32", np.float32).reshape(W + 2, E + 2, H + 2)   # C# float[,,]: z fastest
    blocks = np.fromfile(tmp_path / "blocks.i32", np.int32).reshape(-1, 3)
    counts = np.fromfile(tmp_path / "counts.i32", np.int32)
    verts = np.fromfile(tmp_path / "vertices.f32", np.float32).reshape(-1, 3, 3)
    nrms = np.fromfile(tmp_path / "normals.f32", np.float32).reshape(-1, 3, 3)
    want, offs, _ = oracle_mod.extract_grid(grid, blocks)
    assert len(want) > 1000 and counts.sum() == 3 * len(want)
    assert np.array_equal(np.diff(offs) * 3, counts)
    wv, wn, _ = oracle_mod.bin_triangles(want, len(blocks), voxel_scale=float(scale))   # VoxelTerrain.cs:437-446
    assert np.abs(verts - wv).max() <= 1e-5
    ok = ~np.isnan(wn)
    assert np.array_equal(np.isnan(nrms), ~ok) and np.abs(nrms[ok] - wn[ok]).max() <= 1e-5


@pytest.mark.gpu
def test_host_mirror_device_resident(tmp_path, oracle_mod):
    """_deviceResident = true: Init / Update run on the GPU (vtmc_terrain_*); the grid read back
    equals the oracle's Update bit for bit, the meshes of the last Update match the oracle's
    extraction of that grid."""
    exe = build_host()
    r = subprocess.run([exe, "--gpu-resident", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "HOST-RESIDENT-OK" in r.stdout, r.stdout + r.stderr
    W, E, H, scale, origin = 64, 32, 64, 0.5, (-3.0, 1.0, 2.0)
    ref = oracle_mod.Terrain(W, E, H, scale, origin, seed=4242)
    ref.update([oracle_mod.plane_modifier(6.3, (-100, -100), (100, 100)),
                oracle_mod.sphere_modifier((10.0, 8.0, 15.0), 5.5),
                oracle_mod.cylinder_modifier((2.0, 5.0, 6.0), (1.0, 0.3, 0.5), 20.0, 2.2, add=False),
                oracle_mod.heightmap_modifier(
                    np.array([[4.0 + 0.5 * ((u * 3 + v * 5) % 7) for v in range(9)] for u in range(9)], np.float32),
                    20.0, 24.0, 9.0)])
    dirty = ref.update([oracle_mod.sphere_modifier((5.0, 4.0, 9.0), 2.0, add=False)])
    grid = np.fromfile(tmp_path / "r_grid.f32", np.float32).reshape(W + 2, E + 2, H + 2)
    assert np.array_equal(grid, ref.grid)
    blocks = np.fromfile(tmp_path / "r_blocks.i32", np.int32).reshape(-1, 3)
    assert np.array_equal(blocks, dirty)
    counts = np.fromfile(tmp_path / "r_counts.i32", np.int32)
    verts = np.fromfile(tmp_path / "r_vertices.f32", np.float32).reshape(-1, 3, 3)
    nrms = np.fromfile(tmp_path / "r_normals.f32", np.float32).reshape(-1, 3, 3)
    want, offs, _ = oracle_mod.extract_grid(ref.grid, dirty)
    assert len(want) > 0 and np.array_equal(np.diff(offs) * 3, counts)
    wv, wn, _ = oracle_mod.bin_triangles(want, len(dirty), voxel_scale=scale)
    assert np.abs(verts - wv).max() <= 1e-5 and np.abs(nrms - wn).max() <= 1e-5
