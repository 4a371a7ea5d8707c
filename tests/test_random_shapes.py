"""Randomised parity sweep on the GPU: grid sizes, memory layouts (x fastest, the C# z fastest, padded
rows), field smoothness, dense batches and block lists, soup and indexed output -- every case against
the CPU oracle (cases / offsets / indices bit-exact, floats within 1e-5).  Seeds are fixed: a failure
names its case."""
import os

import numpy as np
import pytest

import fields

# VTMC_FUZZ_SEEDS=200 widens the sweep for a one-off hunt (the default keeps the suite short)
N_SEEDS = int(os.environ.get("VTMC_FUZZ_SEEDS", "12"))

pytestmark = pytest.mark.gpu
ATOL = 1e-5


@pytest.fixture(scope="module")
def ex():
    import volumetricterrain_amd as vt
    e = vt.Extractor(0)
    yield e
    e.close()


def smooth_field(rng, n, order):
    """A few random plane waves: surfaces of varying density, some cells with exact zeros."""
    g = fields._idx((n[0] + 2, n[1] + 2, n[2] + 2), order)
    x, y, z = np.meshgrid(*[np.arange(d + 2, dtype=np.float32) for d in n], indexing="ij")
    acc = np.zeros(x.shape, np.float32)
    for _ in range(rng.integers(1, 5)):
        k = rng.normal(size=3).astype(np.float32) * np.float32(rng.uniform(0.05, 0.9))
        acc += np.float32(rng.uniform(0.3, 1.0)) * np.sin(k[0] * x + k[1] * y + k[2] * z + np.float32(rng.uniform(0, 6.28)))
    acc += np.float32(rng.uniform(-0.5, 0.5))
    if rng.random() < 0.3:
        acc = np.where(rng.random(acc.shape) < 0.05, np.float32(0.0), acc)
    g[...] = acc.astype(np.float32)
    return g


def check(got, want):
    assert len(got) == len(want)
    assert np.array_equal(got["block"], want["block"])
    for f in ("p0", "p1", "p2", "n0", "n1", "n2"):
        nan = np.isnan(want[f])
        assert np.array_equal(np.isnan(got[f]), nan)
        assert np.abs(np.where(nan, 0, got[f]) - np.where(nan, 0, want[f])).max(initial=0.0) <= ATOL


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_grid_soup_and_indexed(ex, oracle_mod, seed):
    rng = np.random.default_rng(1000 + seed)
    n = tuple(int(8 * rng.integers(1, 10)) for _ in range(3))
    order = "x" if rng.random() < 0.6 else "z"
    g = smooth_field(rng, n, order) if rng.random() < 0.7 else fields.random_field(n, seed=seed, order=order)
    blocks = None
    if rng.random() < 0.4:   # a dirty list in arbitrary order, possibly with repeats
        allb = oracle_mod.all_blocks(*n)
        blocks = allb[rng.integers(0, len(allb), size=int(rng.integers(1, len(allb) + 1)))]
    want, want_offs, _ = oracle_mod.extract_grid(g, blocks, threads=4)
    assert ex.extract_grid(g, blocks) == len(want), (seed, n, order)
    got, offs = ex.read_triangles()
    assert np.array_equal(offs, want_offs)
    check(got, want)
    try:
        ex.set_output_mode(True)
        T = ex.extract_grid(g, blocks)
        verts, idx, voffs, toffs = ex.read_indexed_mesh()
    finally:
        ex.set_output_mode(False)
    overts, oidx, ovoffs, otoffs = oracle_mod.extract_grid_indexed(g, blocks)
    assert T == len(want) and np.array_equal(toffs, otoffs) and np.array_equal(voffs, ovoffs)
    assert np.array_equal(idx, oidx), (seed, n, order)
    for f in ("position", "normal"):
        nan = np.isnan(overts[f])
        assert np.abs(np.where(nan, 0, verts[f]) - np.where(nan, 0, overts[f])).max(initial=0.0) <= ATOL


@pytest.mark.parametrize("seed", range(max(4, N_SEEDS // 3)))
def test_random_padded_volume_batches_on_the_device(ex, oracle_mod, seed):
    """Batches of equally shaped volumes with padded row / slab strides handed over as device pointers."""
    import torch
    rng = np.random.default_rng(2000 + seed)
    n = (int(8 * rng.integers(4, 9)), int(8 * rng.integers(1, 5)), int(8 * rng.integers(1, 5)))   # nx >= 32: the streaming classify
    nv = int(rng.integers(1, 5))
    sy = n[0] + 2 + int(rng.integers(0, 7))
    sz = sy * (n[1] + 2) + int(rng.integers(0, 9))
    vs = sz * (n[2] + 2) + int(rng.integers(0, 33))
    host = np.zeros(nv * vs, np.float32)
    want = []
    for v in range(nv):
        g = smooth_field(rng, n, "x")
        view = host[v * vs:v * vs + sz * (n[2] + 2)].reshape(n[2] + 2, sz)[:, :sy * (n[1] + 2)].reshape(n[2] + 2, n[1] + 2, sy)
        view[:, :, :n[0] + 2] = g.transpose(2, 1, 0)
        t, _, _ = oracle_mod.extract_grid(g, threads=4)
        t = t.copy()
        t["block"] += v * (n[0] // 8) * (n[1] // 8) * (n[2] // 8)
        want.append(t)
    want = np.concatenate(want)
    d = torch.from_numpy(host).cuda()
    assert ex.extract_volumes_device(d.data_ptr(), n, (1, sy, sz), nv, vs) == len(want)
    got, _ = ex.read_triangles()
    check(got, want)


@pytest.mark.parametrize("seed", range(max(4, N_SEEDS // 3)))
def test_random_sampled_batches_classify_from_sign_bits(ex, oracle_mod, seed):
    """Device-sampled batches of random shape: the classify stage reading the sampler's sign bits (fill_keeps_signs) and the
    one reading the samples must leave the same block offsets, per-volume counts and bytes; the block offsets are also
    checked against the oracle's count pass over the downloaded field."""
    import torch
    import volumetricterrain_amd as vt
    rng = np.random.default_rng(3000 + seed)
    n = (int(8 * rng.integers(4, 26)), int(8 * rng.integers(1, 7)), int(8 * rng.integers(1, 7)))
    nv = int(rng.integers(1, 5))
    dx, dy, dz = n[0] + 2, n[1] + 2, n[2] + 2
    sv = dx * dy * dz + int(rng.integers(0, 3)) * 64
    org = rng.integers(-200, 200, size=(nv, 3)).astype(np.int32)
    prm = vt.density_params("fbm8" if rng.random() < 0.5 else "perlin3d", int(rng.integers(24, 200)))
    d = torch.empty(nv * sv, dtype=torch.float32, device="cuda")
    bpv = (n[0] // 8) * (n[1] // 8) * (n[2] // 8)
    with vt.Extractor(0) as e2:
        outs = []
        for keep in (0, 1):
            e2.set_tuning(fill_keeps_signs=keep)
            e2.density_fill_device(prm, org, (dx, dy, dz), (1, dx, dx * dy), sv, d.data_ptr())
            T = e2.extract_volumes_device(d.data_ptr(), n, (1, dx, dx * dy), nv, sv)
            _, off_ptr, vc_ptr = e2.device_results()
            outs.append((T, e2.copy_u32(off_ptr, nv * bpv + 1).copy(), e2.copy_u32(vc_ptr, 2 * nv).copy(), e2.read_triangles()[0].tobytes()))
        assert outs[0][0] == outs[1][0], (seed, n, nv)
        assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2]) and outs[0][3] == outs[1][3]
    host = d.cpu().numpy()
    want = [0]
    for v in range(nv):
        g = host[v * sv:v * sv + dx * dy * dz].reshape(dz, dy, dx).transpose(2, 1, 0)
        _, offs, _ = oracle_mod.extract_grid(np.ascontiguousarray(g), threads=4, count_only=True)
        base = want[-1]
        want.extend((np.asarray(offs[1:], np.int64) + base).tolist())
    assert np.array_equal(outs[1][1].astype(np.int64), np.asarray(want, np.int64)), (seed, n, nv)
