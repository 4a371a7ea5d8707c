"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): cube cases, per-block counts/offsets and the triangle -> (block,
cell, table-triangle) structure bit-exact; interpolated positions and normals within 1e-5 absolute.
"""
import numpy as np
import pytest

import fields

pytestmark = pytest.mark.gpu

ATOL = 1e-5  # north_star tolerance for positions / normals


@pytest.fixture
def ex():
    """A fresh context per test: a test that leaves a context in a bad state (or a host-side slip in one auxiliary entry point)
    costs that one test.  What a shared, long-lived context goes through is tests/test_lifecycle.py's subject."""
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import volumetricterrain_amd as vt
    assert vt.library_path() is not None
    e = vt.Extractor(0)
    yield e
    e.close()


def assert_tris_match(got, want, atol=ATOL):
    assert len(got) == len(want)
    assert np.array_equal(got["block"], want["block"])
    worst = 0.0
    for f in ("p0", "p1", "p2", "n0", "n1", "n2"):
        g, w = got[f], want[f]
        nan_g, nan_w = np.isnan(g), np.isnan(w)
        assert np.array_equal(nan_g, nan_w), "NaN pattern differs in " + f
        d = np.abs(np.where(nan_w, 0, g) - np.where(nan_w, 0, w))
        worst = max(worst, float(d.max()) if d.size else 0.0)
    assert worst <= atol, "max abs deviation %g > %g" % (worst, atol)
    return worst


def test_exact_mode_is_bit_compatible_with_oracle(ex, oracle_mod):
    """emit_fast_math=0: correctly rounded divide / sqrt, contraction off => identical floats
    (up to the sign of zero).  The shipped default (v_rcp / v_rsq / fma) stays within ATOL and is
    what every other test runs."""
    g = oracle_mod.density_volume("perlin3d", 64)
    want, want_offs, _ = oracle_mod.extract_grid(g, threads=8)
    try:
        ex.set_tuning(emit_fast_math=0)
        assert ex.extract_grid(g) == len(want)
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, want_offs)
        assert assert_tris_match(got, want, atol=0.0) == 0.0
    finally:
        ex.set_tuning(emit_fast_math=1)
    assert ex.extract_grid(g) == len(want)
    worst = assert_tris_match(ex.read_triangles(False), want)
    assert worst <= 2e-6, worst     # fast path: observed ~5e-7, bar 1e-5


def test_exact_zero_samples_and_unit_weights(ex, oracle_mod):
    """Samples that are exactly 0 give t = -0 / t = 1 (vertices ON lattice points).  A 1-ulp
    reciprocal must not step floor/ceil past the edge (regression: garbage lattice reads turned a
    zero-weight normal into NaN).  The exact-mode kernel runs first so LDS holds foreign leftovers."""
    rng = np.random.default_rng(11)
    n = (32, 32, 32)
    g = fields.random_field(n, seed=3)
    ints = rng.integers(-2, 3, size=g.shape).astype(np.float32)
    g[...] = np.where(rng.random(g.shape) < 0.5, ints, g)      # half the samples are small integers, many 0
    want, want_offs, _ = oracle_mod.extract_grid(g, threads=8)
    try:
        ex.set_tuning(emit_fast_math=0)
        assert ex.extract_grid(g) == len(want)
        assert_tris_match(ex.read_triangles(False), want, atol=0.0)
    finally:
        ex.set_tuning(emit_fast_math=1)
    for _ in range(2):
        assert ex.extract_grid(g) == len(want)
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, want_offs)
        assert_tris_match(got, want)


def test_tile_batch_matches_oracle(ex, oracle_mod):
    """vtmc_extract_blocks on the reference's own tile layout (VoxelTerrain.cs:341-361)."""
    g = oracle_mod.density_volume("perlin3d", 32)
    tiles = oracle_mod.gather_tiles(g, oracle_mod.all_blocks(32, 32, 32))
    tiles = np.concatenate([tiles, fields.all_cases_tile()])
    want, want_offs, want_cases = oracle_mod.extract_tiles(tiles)
    T = ex.extract_blocks(tiles)
    assert T == len(want)
    got, offs = ex.read_triangles()
    assert np.array_equal(offs, want_offs)
    assert np.array_equal(ex.read_cases(), want_cases)
    assert_tris_match(got, want)


@pytest.mark.parametrize("order", ["x", "z"])
def test_grid_in_place_matches_oracle(ex, oracle_mod, order):
    """vtmc_extract_grid, x-fastest (dense streaming classify) and C# z-fastest layouts."""
    g = oracle_mod.density_volume("perlin3d", 64, order=order)
    want, want_offs, want_cases = oracle_mod.extract_grid(g, want_cases=True, threads=8)
    T = ex.extract_grid(g)
    assert T == len(want)
    got, offs = ex.read_triangles()
    assert np.array_equal(offs, want_offs)
    assert np.array_equal(ex.read_cases(), want_cases)
    assert_tris_match(got, want)


def test_non_cubic_and_partial_segments(ex, oracle_mod):
    """nx = 40 / 72 ... exercise partial 64-lane segments of the streaming classify."""
    for n in ((40, 16, 24), (72, 8, 16), (136, 8, 8), (32, 8, 8), (200, 24, 8)):
        g = fields.random_field(n, seed=n[0])
        want, want_offs, _ = oracle_mod.extract_grid(g, threads=8)
        assert ex.extract_grid(g) == len(want)
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, want_offs)
        assert_tris_match(got, want)


def test_queued_extract_regrows_and_equals_blocking(ex, oracle_mod):
    """vtmc_extract_volumes_device_async + vtmc_extract_finish give the bytes the blocking call gives.
    A buffer that is too small on the first launch (reserve 1) exercises the path where the emit
    kernel refuses to run, finish() grows the buffer and queues the emit stage again; the per-volume
    counts are already final behind the queued call (what the all-gather is queued on)."""
    import torch
    c, dim = 64, 66
    chunks = [oracle_mod.density_volume("perlin3d", c, origin=(64 * i, 0, 64)) for i in range(3)]
    d = torch.from_numpy(np.stack([np.ascontiguousarray(g.transpose(2, 1, 0)) for g in chunks])).cuda()
    T0 = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), 3, dim ** 3)
    want, want_offs = ex.read_triangles()
    counts = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
    ex.reserve_triangles(1)
    ex.extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), 3, dim ** 3)
    ex.copy_volume_counts_device(counts.data_ptr(), 3)       # queued behind the scan, before finish
    assert ex.extract_finish() == T0
    got, offs = ex.read_triangles()
    assert got.tobytes() == want.tobytes() and np.array_equal(offs, want_offs)
    torch.cuda.synchronize()
    vc = counts.cpu().numpy()
    assert vc[:, 1].sum() == T0 and np.array_equal(vc[:, 0], 3 * vc[:, 1])
    with pytest.raises(Exception) as e:      # one finish per queued extract
        ex.extract_finish()
    assert e.value.code == -5
    try:                                     # without the events between the kernels: same bytes, only the step's total is timed
        ex.set_tuning(stage_events=0)
        assert ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), 3, dim ** 3) == T0
        assert ex.read_triangles()[0].tobytes() == want.tobytes()
        ms = ex.last_stage_ms()
        assert ms["total"] > 0 and ms["classify"] == 0 and ms["emit"] == 0
    finally:
        ex.set_tuning(stage_events=1)


@pytest.mark.parametrize("dims,n_vol,kind,indexed", [((128, 128, 128), 3, "fbm8", False), ((96, 40, 24), 2, "perlin3d", False),
                                                    ((200, 16, 8), 3, "fbm8", True), ((64, 64, 64), 1, "perlin3d", True)])
def test_classify_from_the_samplers_sign_bits(ex, oracle_mod, dims, n_vol, kind, indexed):
    """tuning key fill_keeps_signs (the streaming driver's setting): the sampler leaves one sign bit per sample and the classify
    stage of the same buffer reads those instead of the samples.  Counts, offsets and output must be what the sample-reading
    classify gives -- partial 64-cell segments, rows that are no multiple of 64 bits, both output modes; a fill of another
    buffer or layout in between must not be mistaken for this one."""
    import torch
    import volumetricterrain_amd as vt
    nx, ny, nz = dims
    sy, sz = nx + 2, (nx + 2) * (ny + 2)
    sv = sz * (nz + 2)
    org = np.array([[37 * i, 5 * i, 11 * i] for i in range(n_vol)], np.int32)
    prm = vt.density_params(kind, 96)

    def results(e):
        T = e.extract_volumes_device(d.data_ptr(), dims, (1, sy, sz), n_vol, sv)
        _, off_ptr, vc_ptr = e.device_results()
        bpv = (nx // 8) * (ny // 8) * (nz // 8)
        out = [T, e.copy_u32(off_ptr, n_vol * bpv + 1).copy(), e.copy_u32(vc_ptr, 2 * n_vol).copy()]
        if indexed:
            v, i, vo, to = e.read_indexed_mesh()
            out += [v.tobytes(), i.tobytes()]
        else:
            out.append(e.read_triangles()[0].tobytes())
        return out

    with vt.Extractor(0) as e2:
        e2.set_output_mode(indexed)
        d = torch.empty(n_vol * sv, dtype=torch.float32, device="cuda")
        other = torch.empty(n_vol * sv, dtype=torch.float32, device="cuda")
        e2.density_fill_device(prm, org, (nx + 2, ny + 2, nz + 2), (1, sy, sz), sv, d.data_ptr())
        want = results(e2)
        assert want[0] > 0
        e2.set_tuning(fill_keeps_signs=1)
        e2.density_fill_device(prm, org, (nx + 2, ny + 2, nz + 2), (1, sy, sz), sv, d.data_ptr())
        got = results(e2)
        assert e2.last_stage_ms()["classify"] > 0
        for a, b in zip(got, want):
            assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
        # signs of another buffer: the extract of `d` falls back to reading samples
        e2.density_fill_device(vt.density_params(kind, 50), org + 3, (nx + 2, ny + 2, nz + 2), (1, sy, sz), sv, other.data_ptr())
        got = results(e2)
        for a, b in zip(got, want):
            assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b


def test_capacity_and_range_errors(ex, oracle_mod):
    """VTMC_ERR_CAPACITY: a destination smaller than T (SURVEY 8b "validate capacity >= T");
    VTMC_ERR_TOO_LARGE: more than 2^31-1 triangles -- the scan's total is kept in 64 bits, so a noise
    field whose count passes 2^32 is refused instead of wrapping to a small count."""
    import ctypes
    import torch
    import volumetricterrain_amd as vt
    g = oracle_mod.density_volume("perlin3d", 32)
    T = ex.extract_grid(g)
    assert T > 100
    buf = np.zeros(T - 1, vt.TRI_DTYPE)
    rc = ex._L.vtmc_read_triangles(ex._h, buf.ctypes.data_as(ctypes.c_void_p), T - 1, None)
    assert rc == -3 and b"capacity" in ex._L.vtmc_last_error(ex._h)
    assert not buf.view(np.uint8).any()                        # nothing was written
    rc = ex._L.vtmc_read_triangles(ex._h, None, T, None)
    assert rc == -1
    # white noise holds 820 / 256 = 3.2 triangles per cell: 1024 x 1024 x 1408 cells -> ~4.7e9 > 2^32
    n = (1024, 1024, 1408)
    gen = torch.Generator(device="cuda").manual_seed(5)
    d = torch.rand((n[2] + 2) * (n[1] + 2) * (n[0] + 2), generator=gen, device="cuda") - 0.5
    ex.reserve_triangles(1024)
    with pytest.raises(vt.VtmcError) as e:
        ex.extract_volumes_device(d.data_ptr(), n, (1, n[0] + 2, (n[0] + 2) * (n[1] + 2)), 1, 0)
    assert e.value.code == -6 and "exceed the int32 range" in str(e.value)
    import re
    reported = int(re.search(r"(\d+) triangles", str(e.value)).group(1))
    assert reported >= 2 ** 32 - 1      # the chained scan saturates its 32-bit triangle words: "at least 2^32 - 1"
    del d
    # the context stays usable
    assert ex.extract_grid(g) == T


def test_dirty_block_list(ex, oracle_mod):
    """Arbitrary, unordered dirty lists (VoxelTerrain.cs:321: hash-set order), both upload paths."""
    g = oracle_mod.density_volume("perlin3d", 64)
    blocks = oracle_mod.all_blocks(64, 64, 64)
    rng = np.random.default_rng(5)
    for count in (7, 300):   # 7: host tile gather path; 300: grid upload + device list
        sel = blocks[rng.permutation(len(blocks))[:count]]
        want, want_offs, _ = oracle_mod.extract_grid(g, sel)
        assert ex.extract_grid(g, sel) == len(want)
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, want_offs)
        assert_tris_match(got, want)


def test_plane_sphere_empty(ex, oracle_mod):
    g = fields.plane((32, 16, 32), 5.375)
    assert ex.extract_grid(g) == 2 * 32 * 32
    tris = ex.read_triangles(False)
    for f in ("n0", "n1", "n2"):
        assert (tris[f] == np.array([0, 1, 0], np.float32)).all()
    for f in ("p0", "p1", "p2"):
        assert (tris[f][:, 1] == np.float32(5.375)).all()
    s = fields.sphere((32, 32, 32), (16.3, 15.6, 16.9), 10.2)
    want, _, _ = oracle_mod.extract_grid(s)
    assert ex.extract_grid(s) == len(want)
    assert_tris_match(ex.read_triangles(False), want)
    for value in (-1.0, 0.0, 2.0, np.nan):
        assert ex.extract_grid(fields.constant((32, 8, 8), value)) == 0
        t, offs = ex.read_triangles()
        assert len(t) == 0 and (offs == 0).all()
        assert (ex.read_cases() == (0xFF if value > 0 else 0)).all()
    assert ex.extract_blocks(np.zeros((0, 1000), np.float32)) == 0


def test_error_behaviour(ex):
    import volumetricterrain_amd as vt
    with pytest.raises(vt.VtmcError) as e:
        ex.extract_grid(np.zeros((12, 10, 10), np.float32))   # 10 cells: not a multiple of 8
    assert e.value.code == -2 and "block size must align" in str(e.value)   # VoxelTerrain.cs:138-139
    with pytest.raises(vt.VtmcError) as e:
        ex.extract_grid(np.zeros((10, 10, 10), np.float32), np.array([[1, 0, 0]], np.int32))
    assert e.value.code == -2


def test_device_volume_batch_matches_grid(ex, oracle_mod):
    """Config 'grid as chunks': 8 chunks of 32^3 with halos in one launch set == per-chunk oracle;
    dense and per-block classify kernels agree."""
    import torch
    n, c = 64, 32
    g = oracle_mod.density_volume("perlin3d", n)
    chunks, want = [], []
    for cz in range(2):
        for cy in range(2):
            for cx in range(2):
                sub = np.ascontiguousarray(
                    g[cx * c:cx * c + c + 2, cy * c:cy * c + c + 2, cz * c:cz * c + c + 2].transpose(2, 1, 0))
                chunks.append(sub)    # memory order z, y, x  => x fastest
                t, _, _ = oracle_mod.extract_grid(sub.transpose(2, 1, 0))
                t = t.copy()
                t["block"] += len(want) * (c // 8) ** 3
                want.append(t)
    want_all = np.concatenate(want)
    d = torch.from_numpy(np.stack(chunks)).cuda()
    dim = c + 2
    for flags in (0, 2):
        T = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), 8, dim ** 3, flags=flags)
        assert T == len(want_all)
        got, offs = ex.read_triangles()
        assert_tris_match(got, want_all)
    # volume_counts = {vertices, triangles} per chunk, the array that gets all-gathered
    _, _, vc_ptr = ex.device_results()
    from volumetricterrain_amd import sharding
    vc = ex.copy_u32(vc_ptr, 16).reshape(8, 2)
    assert np.array_equal(vc[:, 1], [len(w) for w in want])
    assert np.array_equal(vc[:, 0], 3 * vc[:, 1])


class _DeviceArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (no copy, no HIP binding)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def device_view_i32(ptr, count):
    import torch
    return torch.as_tensor(_DeviceArray(ptr, count), device="cuda")


def test_sharded_host_entry(ex, oracle_mod):
    g = oracle_mod.density_volume("perlin3d", 64)
    whole, _, _ = oracle_mod.extract_grid(g, threads=8)
    total = 0
    per_rank = []
    for rank in range(2):
        T, counts = ex.extract_grid_sharded(g, 32, rank, 2)
        assert counts[:, 1].sum() == T and np.array_equal(counts[:, 0], 3 * counts[:, 1])
        per_rank.append(counts)
        total += T
    assert total == len(whole)
    assert len(per_rank[0]) == 4 and len(per_rank[1]) == 4


def test_density_sampler_matches_cpu_twin(ex, oracle_mod):
    """The wave64 column sampler against the per-sample CPU twin (oracle/density_ref.c): x-fastest volumes, the C# z-fastest
    layout and a padded x-fastest volume, several volumes with non-zero origins, 1 / 2 / 3 / 5 / 7 / 8 octaves, and the per-sample kernel that
    serves more than 8 octaves.  The sampler is ONE function of position: whatever the memory layout, the walk is along z with the
    same operation order, so the x-fastest, the z-fastest and the padded fill of the same points agree BIT FOR BIT."""
    import torch
    import volumetricterrain_amd as vt
    for kind in ("perlin3d", "fbm8"):
        n = 64
        dim = n + 2
        want = oracle_mod.density_volume(kind, n)           # [x,y,z], x fastest
        d = torch.empty(dim ** 3, dtype=torch.float32, device="cuda")
        ex.density_fill_device(vt.density_params(kind, n), [[0, 0, 0]], (dim, dim, dim), (1, dim, dim * dim), 0, d.data_ptr())
        got = d.cpu().numpy().reshape(dim, dim, dim).transpose(2, 1, 0).copy()
        assert np.abs(got - want).max() <= 2e-6
        # z fastest (a C# float[,,]): got[x, y, z] directly
        ex.density_fill_device(vt.density_params(kind, n), [[0, 0, 0]], (dim, dim, dim), (dim * dim, dim, 1), 0, d.data_ptr())
        got_z = d.cpu().numpy().reshape(dim, dim, dim)
        assert np.abs(got_z - want).max() <= 2e-6
        assert np.array_equal(got_z.view(np.uint32), got.view(np.uint32)), "x-fastest and z-fastest fills differ in their bits"
        # padded rows and slabs (stride_y > dim_x, stride_z > stride_y * dim_y): still the same bits
        pd = torch.zeros((dim + 3) * (dim + 5) * dim, dtype=torch.float32, device="cuda")
        ex.density_fill_device(vt.density_params(kind, n), [[0, 0, 0]], (dim, dim, dim), (1, dim + 3, (dim + 3) * (dim + 5)), 0, pd.data_ptr())
        got_p = pd.cpu().numpy().reshape(dim, dim + 5, dim + 3)[:, :dim, :dim].transpose(2, 1, 0)
        assert np.array_equal(got_p.view(np.uint32), got.view(np.uint32)), "padded and compact fills differ in their bits"
    # three ragged volumes at non-zero origins, padded rows (stride_y > dim_x)
    dims, pad = (40, 300, 24), 48
    orgs = [(7, 100, 3), (512, 0, 77), (1000, 1700, 1999)]
    for octaves in (1, 2, 3, 5, 7, 8, 11):   # odd counts: the packed walk's last octave pair has an empty half
        prm = vt.density_params("fbm8", 2048)
        prm.octaves = octaves
        oprm = oracle_mod.density_params("fbm8", 2048)
        oprm.octaves = octaves
        vs = pad * dims[1] * dims[2]
        d = torch.zeros(3 * vs, dtype=torch.float32, device="cuda")
        ex.density_fill_device(prm, orgs, dims, (1, pad, pad * dims[1]), vs, d.data_ptr())
        got = d.cpu().numpy().reshape(3, dims[2], dims[1], pad)
        assert not got[..., dims[0]:].any()                  # the padding is never written
        import ctypes
        for v, o in enumerate(orgs):
            want = np.empty((dims[2], dims[1], dims[0]), np.float32)
            oracle_mod.lib().vto_density_fill(ctypes.byref(oprm), o[0], o[1], o[2], dims[0], dims[1], dims[2], 1, dims[0],
                                              dims[0] * dims[1], oracle_mod._p(want))
            assert np.abs(got[v][..., :dims[0]] - want).max() <= 2e-6, (octaves, v)


@pytest.mark.parametrize("dims", [(1, 1, 1), (3, 5, 7), (64, 4, 1), (16, 16, 161), (257, 1, 9), (10, 26, 330)])
def test_density_sampler_small_and_ragged_volumes(ex, oracle_mod, dims):
    """Planes smaller than one workgroup (every lane past the plane's end walks the plane's last column again), a plane of exactly 256 points,
    one of 257, a single step, walks of more than one segment (161, 330 steps): x-fastest with a guard band behind every volume that must stay
    untouched, and the z-fastest fill bit-equal to it.  (The sign words: test_random_sampled_batches_classify_from_sign_bits.)"""
    import ctypes
    import torch
    import volumetricterrain_amd as vt
    dx, dy, dz = dims
    n_vol, guard = 2, 37
    orgs = [(5, 1000, 9), (2040, 3, 1)]
    prm, oprm = vt.density_params("fbm8", 2048), oracle_mod.density_params("fbm8", 2048)
    vs = dx * dy * dz + guard
    d = torch.full((n_vol * vs,), -7.0, dtype=torch.float32, device="cuda")
    ex.density_fill_device(prm, orgs, dims, (1, dx, dx * dy), vs, d.data_ptr())
    got = d.cpu().numpy().reshape(n_vol, vs)
    assert (got[:, dx * dy * dz:] == -7.0).all(), "the sampler wrote behind a volume"
    dzf = torch.full((n_vol * vs,), -7.0, dtype=torch.float32, device="cuda")
    ex.density_fill_device(prm, orgs, dims, (dy * dz, dz, 1), vs, dzf.data_ptr())
    got_z = dzf.cpu().numpy().reshape(n_vol, vs)
    assert (got_z[:, dx * dy * dz:] == -7.0).all()
    for v, o in enumerate(orgs):
        want = np.empty((dz, dy, dx), np.float32)
        oracle_mod.lib().vto_density_fill(ctypes.byref(oprm), o[0], o[1], o[2], dx, dy, dz, 1, dx, dx * dy, oracle_mod._p(want))
        g = got[v, :dx * dy * dz].reshape(dz, dy, dx)
        assert np.abs(g - want).max() <= 2e-6, (dims, v)
        gz = got_z[v, :dx * dy * dz].reshape(dx, dy, dz).transpose(2, 1, 0)
        assert np.array_equal(gz.view(np.uint32), g.view(np.uint32)), "x-fastest and z-fastest fills differ in their bits"


def test_config_256_full_compare(ex, oracle_mod):
    """BASELINE config[1]: 256^3 perlin3d, whole output against the oracle."""
    g = oracle_mod.density_volume("perlin3d", 256)
    want, want_offs, _ = oracle_mod.extract_grid(g, threads=oracle_mod.max_threads())
    assert ex.extract_grid(g) == len(want) == 2655156
    got, offs = ex.read_triangles()
    assert np.array_equal(offs, want_offs)
    assert_tris_match(got, want)


def test_config_1024_chunked_properties(ex, oracle_mod):
    """BASELINE config[2]: 1024^3 as 512 chunks of 128^3 generated on the device.  Size-independent
    checks: per-chunk counts equal the oracle's count pass on a sample of chunks, offsets are a
    valid exclusive scan, every record is well formed, one sampled chunk matches the oracle fully."""
    import torch
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding
    n, c = 1024, 128
    dim = c + 2
    origins = sharding.chunk_origins(n, c)
    d = torch.empty(len(origins) * dim ** 3, dtype=torch.float32, device="cuda")
    ex.density_fill_device(vt.density_params("perlin3d", n), origins, (dim, dim, dim), (1, dim, dim * dim),
                           dim ** 3, d.data_ptr())
    T = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(origins), dim ** 3)
    tri_ptr, off_ptr, vc_ptr = ex.device_results()
    bpv = (c // 8) ** 3
    offs = ex.copy_u32(off_ptr, len(origins) * bpv + 1)
    vc = ex.copy_u32(vc_ptr, 2 * len(origins)).reshape(-1, 2)
    assert offs[0] == 0 and offs[-1] == T and (np.diff(offs.astype(np.int64)) >= 0).all()
    assert vc[:, 1].sum() == T
    assert np.array_equal(np.diff(offs.astype(np.int64))[: bpv * len(origins)].reshape(len(origins), bpv).sum(1), vc[:, 1])
    # EVERY block of EVERY chunk against the oracle's count pass (CollectTriNum.compute:41-64 restated),
    # fed the SAME device-generated array: all 2 097 152 block offsets
    blocks = oracle_mod.all_blocks(c, c, c)
    threads = oracle_mod.max_threads()
    o_offs = np.empty(bpv + 1, np.int32)
    offs64 = offs.astype(np.int64)
    step = 32
    for v0 in range(0, len(origins), step):
        host = d[v0 * dim ** 3:(v0 + step) * dim ** 3].cpu().numpy()
        for v in range(v0, min(v0 + step, len(origins))):
            sub = host[(v - v0) * dim ** 3:(v - v0 + 1) * dim ** 3]
            n_v = oracle_mod.lib().vto_extract_grid(oracle_mod._p(sub), 1, dim, dim * dim, oracle_mod._p(blocks), bpv, None, 0,
                                                    oracle_mod._p(o_offs), None, threads)
            assert n_v == vc[v, 1], "chunk %d: %d triangles, oracle %d" % (v, vc[v, 1], n_v)
            assert np.array_equal(offs64[v * bpv:(v + 1) * bpv + 1] - offs64[v * bpv], o_offs), "chunk %d block offsets" % v
    # sampled chunks in full (positions / normals / block ids)
    for v in (0, 77, 300, 511):
        sub = d[v * dim ** 3:(v + 1) * dim ** 3].cpu().numpy().reshape(dim, dim, dim).transpose(2, 1, 0)
        want, want_offs, _ = oracle_mod.extract_grid(sub, threads=oracle_mod.max_threads())
        assert vc[v, 1] == len(want)
        got = ex.copy_to_host(tri_ptr + 76 * int(offs[v * bpv]), 76 * len(want)).view(vt.TRI_DTYPE)
        want = want.copy()
        want["block"] += v * bpv
        assert_tris_match(got, want)
        assert np.array_equal(offs[v * bpv:(v + 1) * bpv + 1].astype(np.int64) - int(offs[v * bpv]), want_offs)


def test_config_streaming_fbm8_double_buffered(ex, oracle_mod):
    """BASELINE config[4] at test scale: an fbm8 world streamed as double-buffered batches of 128^3
    chunks (batch k+1 sampled while batch k is extracted).  The stream's per-chunk counts equal a
    one-shot extraction of the same chunks, and one chunk of a late batch matches the oracle on the
    device-generated samples."""
    import torch
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding
    from volumetricterrain_amd.streaming import ChunkStream
    world, c = (256, 256, 384), 128
    dim = c + 2
    with ChunkStream(world, chunk=c, batch_chunks=5, kind="fbm8", noise_n=256) as st:   # ramp centre y = 128; 12 chunks -> batches of 5, 5, 2
        assert st.n_batches() == 3
        kept = None
        total, counts = 0, []
        for k, org, T, bex in st.batches():
            tri_ptr, off_ptr, vc_ptr = bex.device_results()
            vc = bex.copy_u32(vc_ptr, 2 * len(org)).reshape(-1, 2)
            counts.append(vc)
            total += T
            if k == 2:   # keep the last chunk of the last batch for the oracle
                offs = bex.copy_u32(off_ptr, len(org) * st.bpv + 1)
                lo, hi = int(offs[(len(org) - 1) * st.bpv]), int(offs[len(org) * st.bpv])
                kept = (org[-1].copy(), bex.copy_to_host(tri_ptr + 76 * lo, 76 * (hi - lo)).view(vt.TRI_DTYPE).copy())
        counts = np.concatenate(counts)
        all_origins = st.origins.copy()
        prm = st.params
    assert counts[:, 1].sum() == total and total > 0
    # one-shot: all 12 chunks generated and extracted in a single batch
    d = torch.empty(len(all_origins) * dim ** 3, dtype=torch.float32, device="cuda")
    ex.density_fill_device(prm, all_origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
    T1 = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(all_origins), dim ** 3)
    _, _, vc_ptr = ex.device_results()
    vc1 = ex.copy_u32(vc_ptr, 2 * len(all_origins)).reshape(-1, 2)
    assert T1 == total and np.array_equal(vc1, counts)
    # the kept chunk against the oracle, fed the device-generated samples
    v = len(all_origins) - 1
    assert np.array_equal(all_origins[v], kept[0])
    sub = d[v * dim ** 3:(v + 1) * dim ** 3].cpu().numpy().reshape(dim, dim, dim).transpose(2, 1, 0)
    want, _, _ = oracle_mod.extract_grid(sub, threads=oracle_mod.max_threads())
    got = kept[1].copy()
    got["block"] %= st.bpv          # batch-local block id = chunk-in-batch * bpv + block
    assert_tris_match(got, want)


def test_max_size_single_grid_equals_chunked(ex, oracle_mod):
    """The largest world the reference accepts (1025 samples per axis, VoxelTerrain.cs:44, plus the
    halo layer) as ONE 1026^3 volume against the same field cut into 512 chunks of 130^3:
    block-decomposition invariance at full size.  Per-chunk triangle counts and
    an order-independent checksum of every record's floats agree; block ids differ by construction."""
    import torch
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import sharding
    n, c = 1024, 128
    prm = vt.density_params("perlin3d", n)
    dim = n + 2
    whole = torch.empty(dim ** 3, dtype=torch.float32, device="cuda")
    ex.density_fill_device(prm, [[0, 0, 0]], (dim, dim, dim), (1, dim, dim * dim), 0, whole.data_ptr())
    T = ex.extract_volumes_device(whole.data_ptr(), (n, n, n), (1, dim, dim * dim), 1, 0)
    tri_ptr, off_ptr, _ = ex.device_results()
    nb = n // 8
    counts = np.diff(ex.copy_u32(off_ptr, nb ** 3 + 1).astype(np.int64)).reshape(nb, nb, nb)   # [bz, by, bx]
    k = c // 8
    per_chunk_whole = counts.reshape(nb // k, k, nb // k, k, nb // k, k).sum(axis=(1, 3, 5)).reshape(-1)     # chunk = cx + ncx*(cy + ncy*cz)

    def float_checksum(ptr, count):
        t = device_view_i32(ptr, count * 19)   # zero-copy view of the library's triangle buffer
        words = t.view(-1, 19)[:, :18].to(torch.int64) & 0xFFFFFFFF
        w = words % 1000003
        return int(words.sum().item()), int((w * w % 1000003).sum().item())

    sum_whole = float_checksum(tri_ptr, T)
    del whole
    cdim = c + 2
    origins = sharding.chunk_origins(n, c)
    d = torch.empty(len(origins) * cdim ** 3, dtype=torch.float32, device="cuda")
    ex.density_fill_device(prm, origins, (cdim, cdim, cdim), (1, cdim, cdim * cdim), cdim ** 3, d.data_ptr())
    T2 = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, cdim, cdim * cdim), len(origins), cdim ** 3)
    tri_ptr, _, vc_ptr = ex.device_results()
    vc = ex.copy_u32(vc_ptr, 2 * len(origins)).reshape(-1, 2)
    # the sampler is one function of position (one walk, one operation order, whatever the layout or the chunking): an exact triangle count.
    # (42 485 756 with the per-sample CPU twin's field: the samples that are zero in exact arithmetic -- the noise lattice points -- sit ~1e-7 off)
    assert T2 == T == 42487270
    assert np.array_equal(vc[:, 1].astype(np.int64), per_chunk_whole)
    assert float_checksum(tri_ptr, T2) == sum_whole


def test_special_values_nan_inf_denormal(ex, oracle_mod):
    """Classification is total: NaN is 'outside' (strict '>', CollectTriNum.compute:50), +inf inside,
    -inf / -0 outside, denormals by sign -- cases, counts and offsets stay bit-exact.  Vertices on an
    edge with a NaN / inf endpoint are unspecified in the reference too (t = -a / (b - a) is NaN and
    the normal fetch goes out of range), so there only the structure is compared; a field of
    denormals, signed zeros and huge finite values is compared in full."""
    rng = np.random.default_rng(21)
    g = fields.random_field((32, 16, 24), seed=9)
    r = rng.random(g.shape)
    g[...] = np.where(r < 0.02, np.nan, g)
    g[...] = np.where((r >= 0.02) & (r < 0.03), np.inf, g)
    g[...] = np.where((r >= 0.03) & (r < 0.04), -np.inf, g)
    n_want, want_offs, want_cases = oracle_mod.extract_grid(g, want_cases=True, count_only=True)
    assert ex.extract_grid(g) == n_want
    got, offs = ex.read_triangles()
    assert np.array_equal(offs, want_offs) and np.array_equal(ex.read_cases(), want_cases)
    assert np.array_equal(got["block"], np.repeat(np.arange(len(offs) - 1, dtype=np.int32), np.diff(offs)))

    h = fields.random_field((32, 16, 24), seed=10)
    h[...] = np.where((r >= 0.04) & (r < 0.07), np.float32(1e-41), h)      # denormal, > 0: inside
    h[...] = np.where((r >= 0.07) & (r < 0.10), np.float32(-1e-41), h)
    h[...] = np.where((r >= 0.10) & (r < 0.13), np.float32(-0.0), h)       # -0 is not > 0: outside
    h[...] = np.where((r >= 0.13) & (r < 0.15), np.float32(3e38), h)
    h[...] = np.where((r >= 0.15) & (r < 0.17), np.float32(-3e38), h)
    want, want_offs, want_cases = oracle_mod.extract_grid(h, want_cases=True, threads=8)
    try:
        ex.set_tuning(emit_fast_math=0)
        assert ex.extract_grid(h) == len(want)
        got, offs = ex.read_triangles()
    finally:
        ex.set_tuning(emit_fast_math=1)
    assert np.array_equal(offs, want_offs) and np.array_equal(ex.read_cases(), want_cases)
    assert np.array_equal(got["block"], want["block"])
    for f in ("p0", "p1", "p2", "n0", "n1", "n2"):   # exact mode: same bits, NaN normals included
        assert np.array_equal(np.isnan(got[f]), np.isnan(want[f]))
        ok = ~np.isnan(want[f])
        assert np.array_equal(got[f][ok], want[f][ok])


def test_streaming_shards_cover_the_world(ex, oracle_mod):
    """ChunkStream with world_size = 2 (both shards run here, one after the other): the two ranks'
    per-chunk counts interleave into exactly the one-rank stream's."""
    from volumetricterrain_amd import sharding
    from volumetricterrain_amd.streaming import ChunkStream
    world, c = (256, 128, 256), 128
    with ChunkStream(world, chunk=c, batch_chunks=3, kind="fbm8", noise_n=128) as st:
        total, counts = st.run()
    parts = []
    for rank in range(2):
        with ChunkStream(world, chunk=c, batch_chunks=1, kind="fbm8", noise_n=128, rank=rank, world_size=2) as st:
            parts.append(st.run())
    merged = sharding.interleave_rank_counts([p[1] for p in parts], 2)
    assert total > 0 and parts[0][0] + parts[1][0] == total
    assert np.array_equal(merged, counts)


@pytest.mark.parametrize("c,n_vol,per_rank,indexed", [(32, 3, 5, False), (128, 2, 2, False), (128, 3, 4, True), (32, 3, 3, True)])
def test_rccl_allgather_of_counts_through_the_c_abi(ex, oracle_mod, c, n_vol, per_rank, indexed):
    """The path's one collective behind the C ABI (vtmc_comm_* + vtmc_allgather_volume_counts): a
    world of one rank round-trips its per-chunk {vertices, triangles} through RCCL, padded to
    volumes_per_rank.  By default the collective is queued behind the emit kernel on the caller's stream; with
    gather_beside = 1 (second round) and 128^3 chunks -- whole scan tiles, the counts leave the scan kernel -- it runs
    on the context's second stream beside the emit kernel.  A second context borrows the communicator
    (vtmc_comm_share: bench.py's two contexts taking turns).  (Two ranks on one device are refused by RCCL; the
    multi-rank path is the same call with world_size > 1 and is covered on CPU by
    tests/test_sharding_gloo.py's layout checks.)"""
    import torch
    import volumetricterrain_amd as vt
    dim = c + 2
    chunks = [oracle_mod.density_volume("perlin3d", c, origin=(c * i, 0, c)) for i in range(n_vol)]
    d = torch.from_numpy(np.stack([np.ascontiguousarray(g.transpose(2, 1, 0)) for g in chunks])).cuda()
    want = [oracle_mod.extract_grid(g, count_only=True)[0] for g in chunks]
    want_v = [len(oracle_mod.extract_grid_indexed(g)[0]) for g in chunks] if indexed else [3 * t for t in want]
    with vt.Extractor(0) as e2:
        e2.set_output_mode(indexed)
        with pytest.raises(vt.VtmcError) as err:
            e2.allgather_volume_counts(0, 4)
        assert err.value.code == -5                       # no communicator yet
        e2.comm_init_rank(e2.comm_unique_id(), 0, 1)
        for rep in range(2):                               # the second round reuses the events and send buffer, beside the emit kernel
            e2.set_tuning(gather_beside=rep)
            gathered = torch.full((1, per_rank, 2), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            e2.extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), n_vol, dim ** 3)
            e2.allgather_volume_counts(gathered.data_ptr(), per_rank)     # queued behind the scan of the pending extract
            host = e2.copy_u32(gathered.data_ptr(), 2 * per_rank).reshape(per_rank, 2)   # on the extract's stream: waits for the collective
            T = e2.extract_finish()
            assert T == sum(want) and list(host[:n_vol, 1]) == want and list(host[:n_vol, 0]) == want_v
            assert not host[n_vol:].any()                  # zero padding up to volumes_per_rank
        # after finish() the same call gathers the finished extract's counts on the caller's stream
        e2.allgather_volume_counts(gathered.data_ptr(), per_rank)
        host = e2.copy_u32(gathered.data_ptr(), 2 * per_rank).reshape(per_rank, 2)
        assert list(host[:n_vol, 1]) == want
        with pytest.raises(vt.VtmcError) as err:
            e2.allgather_volume_counts(gathered.data_ptr(), n_vol - 1)
        assert err.value.code == -3
        # a second context issues its all-gather through the first one's communicator and never destroys it
        with vt.Extractor(0) as e3:
            e3.set_output_mode(indexed)
            with pytest.raises(vt.VtmcError) as err:
                e3.comm_share(e3)
            assert err.value.code == -1
            e3.comm_share(e2)
            gathered.fill_(0x7FFFFFFF)
            torch.cuda.synchronize()
            e3.extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), n_vol, dim ** 3)
            e3.allgather_volume_counts(gathered.data_ptr(), per_rank)
            host = e3.copy_u32(gathered.data_ptr(), 2 * per_rank).reshape(per_rank, 2)
            assert e3.extract_finish() == sum(want) and list(host[:n_vol, 1]) == want and list(host[:n_vol, 0]) == want_v
        # the collective on a stream of the caller's that is NOT the extract's: the library orders it behind the extract's emit launch
        # (bench.py's default at N > 1: the main stream never waits for the collective)
        side = torch.cuda.Stream()
        gathered.fill_(0x7FFFFFFF)
        torch.cuda.synchronize()
        e2.set_tuning(gather_beside=0)
        e2.extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), n_vol, dim ** 3)
        e2.allgather_volume_counts(gathered.data_ptr(), per_rank, side.cuda_stream)
        side.synchronize()
        host = gathered.cpu().numpy().reshape(per_rank, 2)
        assert e2.extract_finish() == sum(want) and list(host[:n_vol, 1]) == want and list(host[:n_vol, 0]) == want_v
        e2.allgather_volume_counts(gathered.data_ptr(), per_rank)      # the borrower is gone, the communicator is not
        host = e2.copy_u32(gathered.data_ptr(), 2 * per_rank).reshape(per_rank, 2)
        assert list(host[:n_vol, 1]) == want
        # bench.py from round 5 on (--streams 2): two contexts take turns, each queues its steps on its own-queue stream; two steps in flight.
        # "own" (the default, --gather-stream main): a step's collective follows its emit kernel on the step's stream -- the collectives of the
        # ONE communicator alternate between two streams and the library chains them by events; "third" (--gather-stream side): every
        # collective on one third stream, the library orders each behind its extract's emit launch.  The copy of the gathered pairs into
        # pinned words follows the collective on its stream.
        with vt.Extractor(0) as e4:
            e4.set_output_mode(indexed)
            e4.comm_share(e2)
            ctxs, third = [e2, e4], torch.cuda.Stream()
            sts = [torch.cuda.ExternalStream(e.stream_handle(own_queue=True)) for e in ctxs]   # each context's stream on a hardware queue of its own
            gs = [torch.full((1, per_rank, 2), 0x7FFFFFFF, dtype=torch.int32, device="cuda") for _ in range(2)]
            hosts = [torch.zeros((per_rank, 2), dtype=torch.int32).pin_memory() for _ in range(2)]
            done = [torch.cuda.Event(), torch.cuda.Event()]
            gathered = [torch.cuda.Event(), torch.cuda.Event()]
            for e in done:
                e.record(third)
            torch.cuda.synchronize()
            for route in ("own", "third", "own"):

                def queue(i):
                    k = i % 2
                    cs = sts[k] if route == "own" else third
                    if route == "own":
                        sts[k].wait_event(done[k])   # the read-back of the step two before (on `third`) has left gs[k]
                    with torch.cuda.stream(cs):
                        gs[k].fill_(0x7FFFFFFF)     # in front of this step's collective, behind the copy of the step two before
                    ctxs[k].extract_volumes_device_async(d.data_ptr(), (c, c, c), (1, dim, dim * dim), n_vol, dim ** 3, sts[k].cuda_stream)
                    ctxs[k].allgather_volume_counts(gs[k].data_ptr(), per_rank, cs.cuda_stream)
                    # the pinned read-back never rides an own-queue stream ("Streams"): behind the collective's event on the ordinary one
                    gathered[k].record(cs)
                    third.wait_event(gathered[k])
                    with torch.cuda.stream(third):
                        hosts[k].copy_(gs[k].view(per_rank, 2), non_blocking=True)
                    done[k].record(third)

                def take(i):
                    k = i % 2
                    done[k].synchronize()
                    assert ctxs[k].extract_finish() == sum(want)
                    h = hosts[k].numpy()
                    assert list(h[:n_vol, 1]) == want and list(h[:n_vol, 0]) == want_v and not h[n_vol:].any()

                for i in range(6):
                    queue(i)
                    if i >= 1:
                        take(i - 1)
                take(5)
                torch.cuda.synchronize()
            # nothing of torch's may outlive the contexts' streams: PyTorch records an event on every stream a pinned tensor was copied on WHEN IT
            # FREES the tensor, and events / ExternalStream wrappers hold the raw handle
            del queue, take, sts, done, gathered, hosts, gs
            import gc
            gc.collect()
            torch.cuda.synchronize()
        e2.comm_destroy()


@pytest.mark.parametrize("indexed", [False, True])
def test_chunk_file_written_and_reloaded_on_the_device(ex, oracle_mod, tmp_path, indexed):
    """SURVEY 8f rank 4 on the device: extract a batch, persist one chunk (image packed by device
    kernels), reload it in another context, re-extract from the reloaded samples: bit-identical mesh.
    The file is also read by the host-side reader (chunkfile.py) and compared with the oracle."""
    import torch
    import volumetricterrain_amd as vt
    from volumetricterrain_amd import chunkfile
    c, dim, bpv = 32, 34, 64
    orgs = [(0, 0, 0), (32, 0, 0), (0, 32, 64)]
    chunks = [oracle_mod.density_volume("perlin3d", 128, origin=o, dims=(dim, dim, dim)) for o in orgs]
    d = torch.from_numpy(np.stack([np.ascontiguousarray(g.transpose(2, 1, 0)) for g in chunks])).cuda()
    path = tmp_path / "chunk2.vtchunk"
    try:
        ex.set_output_mode(indexed)
        ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), 3, dim ** 3)
        ex.chunk_write(path, 2, orgs[2], with_samples=True)
        f = chunkfile.read_chunk(path)
        assert f["origin"] == orgs[2] and f["cells"] == (c, c, c) and f["flags"] == (1 | (4 if indexed else 2))
        assert np.array_equal(f["samples"], np.ascontiguousarray(chunks[2].transpose(2, 1, 0)).ravel())
        if indexed:
            verts, idx, voffs, toffs = oracle_mod.extract_grid_indexed(chunks[2])
            assert np.array_equal(f["tri_offsets"], toffs.astype(np.uint32)) and np.array_equal(f["vert_offsets"], voffs.astype(np.uint32))
            assert np.array_equal(f["indices"], idx)
            assert np.abs(f["vertices"]["position"] - verts["position"]).max() <= ATOL
            assert np.abs(f["vertices"]["normal"] - verts["normal"]).max() <= ATOL
        else:
            want, want_offs, _ = oracle_mod.extract_grid(chunks[2])
            assert np.array_equal(f["tri_offsets"], want_offs.astype(np.uint32))
            assert_tris_match(f["triangles"], want)            # `block` is chunk-relative in the file
        with vt.Extractor(0) as e2:
            e2.set_output_mode(indexed)
            view = e2.chunk_read(path)
            assert (view.n_blocks, view.n_triangles) == (bpv, len(f["indices"]) if indexed else len(f["triangles"]))
            T = e2.extract_volumes_device(view.d_samples, (c, c, c), (1, dim, dim * dim), 1, 0)
            assert T == view.n_triangles
            if indexed:
                v2, i2, vo2, to2 = e2.read_indexed_mesh()
                assert v2.tobytes() == f["vertices"].tobytes() and np.array_equal(i2, f["indices"])
                assert np.array_equal(vo2, f["vert_offsets"]) and np.array_equal(to2, f["tri_offsets"])
                assert e2.copy_to_host(view.d_vertices, 24 * view.n_vertices).tobytes() == f["vertices"].tobytes()
            else:
                t2, o2 = e2.read_triangles()
                assert t2.tobytes() == f["triangles"].tobytes() and np.array_equal(o2, f["tri_offsets"])
                assert e2.copy_to_host(view.d_triangles, 76 * view.n_triangles).tobytes() == f["triangles"].tobytes()
            (tmp_path / "bad").write_bytes(b"VTCHUNK1" + b"\0" * 40)
            with pytest.raises(vt.VtmcError) as err:
                e2.chunk_read(tmp_path / "bad")
            assert err.value.code == -1
            # hostile headers: counts far beyond the file (must be refused before anything is allocated), cells beyond the
            # format's limit, unknown flags, a truncated body -- VTMC_ERR_INVALID_ARG each time, the process stays alive
            good = bytearray(open(path, "rb").read())
            import struct

            def patched(offset, fmt, *values):
                b = bytearray(good)
                struct.pack_into(fmt, b, offset, *values)
                return bytes(b)

            hostile = {
                "huge_triangles": patched(44, "<I", 0xFFFFFFF0),                 # n_triangles
                "huge_vertices": patched(48, "<I", 0xFFFFFFF0),                  # n_vertices
                "huge_cells": patched(28, "<iii", 0x7FFFFFF8, 8, 8),             # cells: + 2 would overflow int32
                "blocks_mismatch": patched(40, "<I", bpv + 1),
                "unknown_flags": patched(12, "<I", 0x80000000 | struct.unpack_from("<I", good, 12)[0]),
                "truncated": bytes(good[:len(good) - 16]),
                "trailing_bytes": bytes(good) + b"\0" * 16,
            }
            for name, blob in hostile.items():
                (tmp_path / name).write_bytes(blob)
                with pytest.raises(vt.VtmcError) as err:
                    e2.chunk_read(tmp_path / name)
                assert err.value.code == -1, name
            assert e2.chunk_read(path).n_triangles == view.n_triangles   # and the good file still loads
    finally:
        ex.set_output_mode(False)
