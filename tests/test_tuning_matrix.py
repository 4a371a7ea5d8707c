"""Every kernel variant vtmc_set_tuning can select, against the CPU oracle.

vtmc_set_tuning is public ABI (include/vtmc.h): a key a host can set is product code.  The product library accepts exactly the keys
below (the *_ablate diagnostics exist in -DVTMC_DIAGNOSTICS builds only and are refused here); each value set runs three inputs --
a Perlin volume through the streaming classify, a random field whose blocks hold far more than 384 triangles / 192 vertices (the
spill paths of the vertex-once and indexed kernels) and the four all-256-cases tiles through the tile route -- in the reference's
76-byte format and, where the key applies to it, in the indexed format.  Bar: offsets / indices bit-exact, floats within 1e-5
(exact mode: equal bits)."""
import threading

import numpy as np
import pytest

import fields
from test_gpu_parity import assert_tris_match

pytestmark = pytest.mark.gpu

DEFAULTS = dict(emit_fast_math=1, emit_once=1, emit_dynamic=1, emit_sub_log2=1, emit_row_masks=1, emit_wgs_per_cu=0,
                classify_wgs_per_cu=3, stage_events=1, gather_beside=0, fill_keeps_signs=0, density_wgs_per_cu=0, place_outputs=0)

# (tuning, runs in soup mode, runs in indexed mode)
SETS = [
    (dict(), True, True),
    (dict(emit_once=0), True, False),
    (dict(emit_fast_math=0), True, True),
    (dict(emit_fast_math=0, emit_once=0), True, False),
    (dict(emit_dynamic=0), True, True),
    (dict(emit_dynamic=0, emit_once=0), True, False),
    (dict(emit_row_masks=0), True, True),
    (dict(emit_row_masks=0, emit_once=0), True, False),
    (dict(emit_sub_log2=0), True, True),
    (dict(emit_sub_log2=3), True, True),
    (dict(emit_sub_log2=4, emit_once=0), True, False),
    (dict(emit_wgs_per_cu=1), True, True),
    (dict(emit_wgs_per_cu=2), True, True),
    (dict(emit_wgs_per_cu=3, emit_once=0), True, False),
    (dict(emit_wgs_per_cu=8), True, True),          # more than fit: the launch still covers the list
    (dict(emit_dynamic=0, emit_row_masks=0), False, True),
    (dict(classify_wgs_per_cu=0), True, True),
    (dict(classify_wgs_per_cu=2), True, True),
    (dict(classify_wgs_per_cu=4), True, False),
    (dict(classify_wgs_per_cu=7), True, False),
    (dict(stage_events=0), True, True),
    # round 6: output placement trials (the emit stage run into several allocations, the fastest kept): every candidate holds the complete result
    (dict(place_outputs=2), True, True),
    (dict(place_outputs=4, emit_once=0), True, False),
    (dict(place_outputs=8, stage_events=0), True, True),
    (dict(place_outputs=16), True, False),
]


@pytest.fixture(scope="module")
def ex():
    import torch
    assert torch.cuda.is_available()
    import volumetricterrain_amd as vt
    e = vt.Extractor(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def cases(oracle_mod):
    """inputs and the oracle's answers, computed once"""
    out = []
    perlin = oracle_mod.density_volume("perlin3d", 64)
    dense = fields.random_field((40, 16, 24), seed=5)      # 40 cells along x: a partial 64-lane segment; blocks of ~1300 triangles
    wide_x = fields.random_field((200, 16, 8), seed=9, scale=1.0)        # 200 cells along x: three 64-cell segments of the streaming classify + a partial one
    wide_z = fields.random_field((16, 8, 136), seed=10, order="z")         # the C# z-fastest order, 136 cells along z
    smooth = oracle_mod.density_volume("perlin3d", 1024, origin=(300, 40, 900), dims=(258, 18, 10))   # 256 cells along x, mostly empty bricks
    for name, g in (("perlin64", perlin), ("random40x16x24", dense), ("random200x16x8", wide_x), ("random16x8x136_zfast", wide_z), ("perlin256x16x8", smooth)):
        soup, offs, _ = oracle_mod.extract_grid(g, threads=8)
        out.append(dict(name=name, grid=g, tiles=None, soup=soup, offs=offs, indexed=oracle_mod.extract_grid_indexed(g)))
    assert np.diff(out[1]["offs"]).max() > 384 and np.diff(out[1]["indexed"][2]).max() > 255
    tiles = fields.all_cases_tile()
    soup, offs, _ = oracle_mod.extract_tiles(tiles)
    out.append(dict(name="all_cases_tiles", grid=None, tiles=tiles, soup=soup, offs=offs, indexed=None))
    return out


def run_case(ex, c, indexed, exact):
    if c["grid"] is not None:
        T = ex.extract_grid(c["grid"])
    else:
        T = ex.extract_blocks(c["tiles"])
    if not indexed:
        assert T == len(c["soup"]), c["name"]
        got, offs = ex.read_triangles()
        assert np.array_equal(offs, c["offs"]), c["name"]
        assert_tris_match(got, c["soup"], atol=0.0 if exact else 1e-5)
        return
    want_v, want_i, want_vo, want_to = c["indexed"]
    assert T == len(want_i), c["name"]
    verts, idx, voffs, toffs = ex.read_indexed_mesh()
    assert np.array_equal(voffs, want_vo) and np.array_equal(toffs, want_to) and np.array_equal(idx, want_i), c["name"]
    for f in ("position", "normal"):
        nan_w = np.isnan(want_v[f])
        assert np.array_equal(np.isnan(verts[f]), nan_w)
        d = np.abs(np.where(nan_w, 0, verts[f]) - np.where(nan_w, 0, want_v[f]))
        assert (float(d.max()) if d.size else 0.0) <= (0.0 if exact else 1e-5), (c["name"], f)


@pytest.mark.parametrize("tuning,soup,indexed", SETS, ids=[",".join("%s=%d" % kv for kv in s[0].items()) or "defaults" for s in SETS])
def test_variant_matches_oracle(ex, cases, tuning, soup, indexed):
    try:
        ex.set_tuning(**dict(DEFAULTS, **tuning))
        exact = tuning.get("emit_fast_math", 1) == 0
        for mode in ([False] if soup else []) + ([True] if indexed else []):
            ex.set_output_mode(mode)
            for c in cases:
                if mode and c["indexed"] is None:
                    continue
                run_case(ex, c, mode, exact)
    finally:
        ex.set_output_mode(False)
        ex.set_tuning(**DEFAULTS)


def test_sampler_residency_cap_matches_cpu_twin(ex, oracle_mod):
    """density_wgs_per_cu (what streaming.ChunkStream's sampler_wgs_per_cu sets): the capped launch writes the same bits as the default
    one, and both sit within 2e-6 of the per-sample CPU twin."""
    import torch
    import volumetricterrain_amd as vt
    n, dim = 64, 66
    want = oracle_mod.density_volume("fbm8", n)
    d = torch.empty(dim ** 3, dtype=torch.float32, device="cuda")
    got = {}
    try:
        for cap in (0, 2, 3):
            ex.set_tuning(density_wgs_per_cu=cap)
            ex.density_fill_device(vt.density_params("fbm8", n), [[0, 0, 0]], (dim, dim, dim), (1, dim, dim * dim), 0, d.data_ptr())
            got[cap] = d.cpu().numpy().reshape(dim, dim, dim).transpose(2, 1, 0).copy()
            assert np.abs(got[cap] - want).max() <= 2e-6
            assert np.array_equal(got[cap].view(np.uint32), got[0].view(np.uint32))
    finally:
        ex.set_tuning(density_wgs_per_cu=0)


def test_refused_keys_and_values(ex):
    """Keys whose code left the product (one_pass*, emit_async, emit_group_log2, emit_idx_waves), the diagnostic *_ablate keys and values outside a
    key's range answer VTMC_ERR_INVALID_ARG and change nothing."""
    import volumetricterrain_amd as vt
    for key, value in (("one_pass", 1), ("one_pass_depth", 2), ("one_pass_unit", 1), ("one_pass_prefetch", 1), ("emit_async", 0),
                       ("emit_group_log2", 2), ("emit_ablate", 1), ("classify_ablate", 1), ("density_ablate", 1), ("no_such_key", 0),
                       ("emit_idx_waves", 3), ("emit_idx_waves", 4), ("classify_wide", 1), ("emit_sub_log2", 5), ("emit_sub_log2", -1), ("emit_wgs_per_cu", 9),
                       ("classify_wgs_per_cu", 8), ("classify_wgs_per_cu", 1), ("density_wgs_per_cu", 1), ("density_wgs_per_cu", 4), ("emit_fast_math", 2), ("emit_once", -1), ("place_outputs", 17), ("place_outputs", -1),
                       ("classify_column", 4), ("classify_column_wgs", 2)):
        with pytest.raises(vt.VtmcError) as e:
            ex.set_tuning(**{key: value})
        assert e.value.code == -1, key   # VTMC_ERR_INVALID_ARG
    g = fields.sphere((16, 16, 16), (8.3, 8.1, 7.9), 5.2)
    assert ex.extract_grid(g) > 0       # the context is as usable as before


def test_two_contexts_on_two_host_threads(oracle_mod):
    """include/vtmc.h "Threading": different contexts may be used from different threads (the reference itself only calls from the Unity
    main thread, TerrainEngine.cs:145-149).  Two host threads, one context (and stream) each, extract different grids concurrently --
    soup on one, indexed and exact-mode soup on the other, dirty lists and whole grids -- every result against the oracle."""
    import volumetricterrain_amd as vt
    ga = oracle_mod.density_volume("perlin3d", 64)
    gb = oracle_mod.density_volume("fbm8", 64, origin=(64, 0, 128))
    want_a = oracle_mod.extract_grid(ga, threads=4)
    want_b = oracle_mod.extract_grid(gb, threads=4)
    want_bi = oracle_mod.extract_grid_indexed(gb)
    blocks = oracle_mod.all_blocks(64, 64, 64)[5::7]
    want_al = oracle_mod.extract_grid(ga, blocks, threads=4)
    errors = []
    start = threading.Barrier(2)

    def worker_a():
        try:
            e = vt.Extractor(0)
            start.wait()
            for it in range(12):
                if it % 3 == 2:
                    assert e.extract_grid(ga, blocks) == len(want_al[0])
                    got, offs = e.read_triangles()
                    assert np.array_equal(offs, want_al[1])
                    assert_tris_match(got, want_al[0])
                else:
                    assert e.extract_grid(ga) == len(want_a[0])
                    got, offs = e.read_triangles()
                    assert np.array_equal(offs, want_a[1])
                    assert_tris_match(got, want_a[0])
            e.close()
        except BaseException as exc:   # noqa: BLE001 -- handed to the main thread
            errors.append(("a", exc))

    def worker_b():
        try:
            e = vt.Extractor(0)
            start.wait()
            for it in range(12):
                if it % 2:
                    e.set_output_mode(True)
                    assert e.extract_grid(gb) == len(want_bi[1])
                    verts, idx, voffs, toffs = e.read_indexed_mesh()
                    assert np.array_equal(idx, want_bi[1]) and np.array_equal(voffs, want_bi[2]) and np.array_equal(toffs, want_bi[3])
                    assert np.abs(verts["position"] - want_bi[0]["position"]).max() <= 1e-5
                else:
                    e.set_output_mode(False)
                    e.set_tuning(emit_fast_math=0)
                    assert e.extract_grid(gb) == len(want_b[0])
                    got, offs = e.read_triangles()
                    assert np.array_equal(offs, want_b[1])
                    assert_tris_match(got, want_b[0], atol=0.0)
                    e.set_tuning(emit_fast_math=1)
            e.close()
        except BaseException as exc:   # noqa: BLE001
            errors.append(("b", exc))

    ts = [threading.Thread(target=worker_a), threading.Thread(target=worker_b)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a worker thread hung"
    assert not errors, errors
