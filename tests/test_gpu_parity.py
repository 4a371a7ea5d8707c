"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): cube cases, per-block counts/offsets and the triangle -> (block,
cell, table-triangle) structure bit-exact; interpolated positions and normals within 1e-5 absolute.
"""
import numpy as np
import pytest

import fields

pytestmark = pytest.mark.gpu

ATOL = 1e-5  # north_star tolerance for positions / normals
SWEEP_DEFAULT = 0  # shipped pipeline: classify -> scan -> emit (sweep=1: the single-pass kernel)


@pytest.fixture(scope="module")
def ex():
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
    (up to the sign of zero), in both pipelines.  The shipped default (v_rcp / v_rsq / fma) stays
    within ATOL and is what every other test runs."""
    g = oracle_mod.density_volume("perlin3d", 64)
    want, want_offs, _ = oracle_mod.extract_grid(g, threads=8)
    try:
        for sweep in (0, 1):
            ex.set_tuning(emit_fast_math=0, sweep=sweep)
            assert ex.extract_grid(g) == len(want)
            got, offs = ex.read_triangles()
            assert np.array_equal(offs, want_offs)
            assert assert_tris_match(got, want, atol=0.0) == 0.0
    finally:
        ex.set_tuning(emit_fast_math=1, sweep=SWEEP_DEFAULT)
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


@pytest.mark.parametrize("sweep", [1, 0])
def test_non_cubic_and_partial_segments(ex, oracle_mod, sweep):
    """nx = 40 / 72 exercise partial 64-lane segments of the streaming classify; sweep=1 is the
    single-pass kernel (default), sweep=0 the classify -> scan -> emit kernels."""
    try:
        ex.set_tuning(sweep=sweep)
        for n in ((40, 16, 24), (72, 8, 16), (136, 8, 8), (32, 8, 8), (200, 24, 8)):
            g = fields.random_field(n, seed=n[0])
            want, want_offs, _ = oracle_mod.extract_grid(g, threads=8)
            assert ex.extract_grid(g) == len(want)
            got, offs = ex.read_triangles()
            assert np.array_equal(offs, want_offs)
            assert_tris_match(got, want)
    finally:
        ex.set_tuning(sweep=SWEEP_DEFAULT)


def test_sweep_and_staged_pipelines_agree_bitwise(ex, oracle_mod):
    """The single-pass kernel and the three-stage pipeline share the emit routine: same bytes out.
    A buffer that is too small on the first call (reserve 1) exercises the count-then-regrow path."""
    g = oracle_mod.density_volume("perlin3d", 128)
    outs = []
    try:
        for sweep in (1, 0):
            ex.set_tuning(sweep=sweep)
            ex.reserve_triangles(1)
            T = ex.extract_grid(g)
            tris, offs = ex.read_triangles()
            outs.append((T, tris.tobytes(), offs.tobytes()))
    finally:
        ex.set_tuning(sweep=SWEEP_DEFAULT)
    assert outs[0] == outs[1]


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
    vc = sharding.copy_device_u32(vc_ptr, 16).reshape(8, 2)
    assert np.array_equal(vc[:, 1], [len(w) for w in want])
    assert np.array_equal(vc[:, 0], 3 * vc[:, 1])


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
    import torch
    import volumetricterrain_amd as vt
    for kind in ("perlin3d", "fbm8"):
        n = 64
        want = oracle_mod.density_volume(kind, n)           # [x,y,z], x fastest
        d = torch.empty((n + 2) ** 3, dtype=torch.float32, device="cuda")
        dim = n + 2
        ex.density_fill_device(vt.density_params(kind, n), [[0, 0, 0]], (dim, dim, dim), (1, dim, dim * dim), 0,
                               d.data_ptr())
        got = d.cpu().numpy().reshape(dim, dim, dim).transpose(2, 1, 0)
        assert np.abs(got - want).max() <= 2e-6


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
    offs = sharding.copy_device_u32(off_ptr, len(origins) * bpv + 1)
    vc = sharding.copy_device_u32(vc_ptr, 2 * len(origins)).reshape(-1, 2)
    assert offs[0] == 0 and offs[-1] == T and (np.diff(offs.astype(np.int64)) >= 0).all()
    assert vc[:, 1].sum() == T
    assert np.array_equal(np.diff(offs.astype(np.int64))[: bpv * len(origins)].reshape(len(origins), bpv).sum(1), vc[:, 1])
    # sampled chunks against the oracle, fed the SAME device-generated array
    for v in (0, 77, 300, 511):
        sub = d[v * dim ** 3:(v + 1) * dim ** 3].cpu().numpy().reshape(dim, dim, dim).transpose(2, 1, 0)
        want, want_offs, _ = oracle_mod.extract_grid(sub, threads=oracle_mod.max_threads())
        assert vc[v, 1] == len(want)
        got = sharding.copy_device_bytes(tri_ptr + 76 * int(offs[v * bpv]), 76 * len(want)).view(vt.TRI_DTYPE)
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
            vc = sharding.copy_device_u32(vc_ptr, 2 * len(org)).reshape(-1, 2)
            counts.append(vc)
            total += T
            if k == 2:   # keep the last chunk of the last batch for the oracle
                offs = sharding.copy_device_u32(off_ptr, len(org) * st.bpv + 1)
                lo, hi = int(offs[(len(org) - 1) * st.bpv]), int(offs[len(org) * st.bpv])
                kept = (org[-1].copy(), sharding.copy_device_bytes(tri_ptr + 76 * lo, 76 * (hi - lo)).view(vt.TRI_DTYPE).copy())
        counts = np.concatenate(counts)
        all_origins = st.origins.copy()
        prm = st.params
    assert counts[:, 1].sum() == total and total > 0
    # one-shot: all 12 chunks generated and extracted in a single batch
    d = torch.empty(len(all_origins) * dim ** 3, dtype=torch.float32, device="cuda")
    ex.density_fill_device(prm, all_origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
    T1 = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(all_origins), dim ** 3)
    _, _, vc_ptr = ex.device_results()
    vc1 = sharding.copy_device_u32(vc_ptr, 2 * len(all_origins)).reshape(-1, 2)
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
    counts = np.diff(sharding.copy_device_u32(off_ptr, nb ** 3 + 1).astype(np.int64)).reshape(nb, nb, nb)   # [bz, by, bx]
    k = c // 8
    per_chunk_whole = counts.reshape(nb // k, k, nb // k, k, nb // k, k).sum(axis=(1, 3, 5)).reshape(-1)     # chunk = cx + ncx*(cy + ncy*cz)

    def float_checksum(ptr, count):
        t = torch.empty(count * 19, dtype=torch.int32, device="cuda")
        sharding._hiprt().hipMemcpy(t.data_ptr(), ptr, 76 * count, 3)   # device -> device
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
    vc = sharding.copy_device_u32(vc_ptr, 2 * len(origins)).reshape(-1, 2)
    assert T2 == T == 42485756
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
