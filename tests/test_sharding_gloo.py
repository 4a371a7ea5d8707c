"""N > 1 path on CPU: world_size-2 `gloo` run of the sharding logic (chunk ownership, the single
all-gather of per-chunk {vertices, triangles}, global offsets).  Per-chunk counts come from the CPU
oracle here (no GPU in this container); on the GPU box the same code all-gathers the counts the
HIP path produced (bench.py, backend 'nccl' = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, chunk, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import oracle
    from volumetricterrain_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    origins = sharding.chunk_origins(n, chunk, rank, world)
    dim = chunk + 2
    local = []
    for o in origins:
        g = oracle.density_volume("perlin3d", n, origin=tuple(int(v) for v in o), dims=(dim, dim, dim))
        T, _, _ = oracle.extract_grid(g, count_only=True)
        local.append((3 * T, T))
    local = torch.tensor(local, dtype=torch.int32).reshape(-1, 2)
    gathered = sharding.allgather_counts(local)                       # (world, n_local, 2)
    per_rank = [gathered[r].numpy() for r in range(world)]
    counts = sharding.interleave_rank_counts(per_rank, world)          # global chunk order
    offs = sharding.global_offsets(counts)
    np.save(os.path.join(out_dir, "offs_%d.npy" % rank), offs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allgather_of_chunk_counts(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    from volumetricterrain_amd import sharding
    n, chunk, world = 64, 32, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, chunk, str(tmp_path)), nprocs=world, join=True)
    # single-process truth: every chunk of the whole grid, in chunk order
    want = []
    for o in sharding.chunk_origins(n, chunk):
        g = oracle_mod.density_volume("perlin3d", n, origin=tuple(int(v) for v in o), dims=(chunk + 2,) * 3)
        T, _, _ = oracle_mod.extract_grid(g, count_only=True)
        want.append((3 * T, T))
    want_offs = sharding.global_offsets(np.array(want))
    whole, _, _ = oracle_mod.extract_grid(oracle_mod.density_volume("perlin3d", n), count_only=True)
    assert want_offs[-1, 1] == whole                                   # chunking keeps every triangle
    for r in range(world):
        got = np.load(tmp_path / ("offs_%d.npy" % r))
        assert np.array_equal(got, want_offs)                          # every rank derives the same offsets


def test_chunk_ownership_is_a_partition():
    from volumetricterrain_amd import sharding
    n, chunk = (64, 32, 96), 32
    all_o = sharding.chunk_origins(n, chunk)
    assert len(all_o) == 2 * 1 * 3
    for world in (1, 2, 3, 4):
        parts = [sharding.chunk_origins(n, chunk, r, world) for r in range(world)]
        merged = np.zeros_like(all_o)
        for r, p in enumerate(parts):
            merged[r::world] = p
        assert np.array_equal(merged, all_o)
    with pytest.raises(ValueError):
        sharding.chunk_grid(100, 32)


def test_balanced_assignment_is_a_partition_evener_than_modulo_and_the_same_on_every_rank(oracle_mod):
    """Round 6: the chunks of a strong-scaling run are cut again by the counts every rank holds after the first all-gather.  512 chunks of a
    perlin3d world (256^3 as 32^3-cell chunks here), counts from the oracle: a partition with equal chunk numbers, never less even than c % N,
    a pure function of the counts (so every rank derives the same lists), and the slot permutation puts every chunk where its rank sent it."""
    from volumetricterrain_amd import sharding
    n, chunk = 256, 32
    costs = []
    for o in sharding.chunk_origins(n, chunk):
        g = oracle_mod.density_volume("perlin3d", n, origin=tuple(int(v) for v in o), dims=(chunk + 2,) * 3)
        costs.append(oracle_mod.extract_grid(g, count_only=True)[0])
    costs = np.array(costs)
    assert len(costs) == 512 and costs.sum() > 0
    for world in (2, 4, 8, 3):
        a = sharding.balanced_assignment(costs, world)
        assert sorted(c for part in a for c in part) == list(range(512))
        cap = -(-512 // world)
        assert all(len(part) <= cap for part in a) and all(part == sorted(part) for part in a)
        assert a == sharding.balanced_assignment(list(costs), world)
        m, b = sharding.imbalance(costs, sharding.modulo_assignment(512, world)), sharding.imbalance(costs, a)
        assert 1.0 <= b <= m + 1e-12
        if world == 8:
            assert b < 1.005 < m          # c % 8 leaves percents on the table, the greedy cut per mills
        perm = sharding.slot_permutation(a, cap)
        gathered = np.zeros(world * cap, np.int64)
        for r, part in enumerate(a):
            gathered[r * cap:r * cap + len(part)] = costs[part]
        assert np.array_equal(gathered[perm], costs)
    assert np.array_equal(sharding.origins_of(n, chunk, range(512)), sharding.chunk_origins(n, chunk))
    assert np.array_equal(sharding.origins_of(n, chunk, sharding.owned_chunks(512, 3, 8)), sharding.chunk_origins(n, chunk, 3, 8))


def test_stream_cpu_sample_is_stratified_by_triangle_counts():
    """bench.py's pick for the CPU leg of config 5: a terrain world has its surface in a fifth of its chunks; an evenly spread sample of 8 of 4096
    missed it altogether in round 5 (triangles_in_sample = 0).  The pick must contain surface chunks and match the world's triangles per chunk."""
    import bench
    rng = np.random.default_rng(7)
    counts = np.zeros(4096, np.int64)
    surface = rng.choice(4096, 600, replace=False)
    counts[surface] = rng.integers(40000, 260000, len(surface))
    for n_sample in (8, 16, 5):
        idx = bench.pick_representative_chunks(counts, n_sample)
        assert len(idx) == n_sample and len(set(idx)) == n_sample and all(0 <= i < 4096 for i in idx)
        ratio = counts[idx].mean() / counts.mean()
        assert counts[idx].sum() > 0 and 0.8 < ratio < 1.2, ratio
    assert bench.pick_representative_chunks(np.zeros(10, np.int64), 4) and len(bench.pick_representative_chunks(counts[:3], 8)) == 3
    full = np.full(64, 1000, np.int64)
    assert len(bench.pick_representative_chunks(full, 8)) == 8      # no empty chunks at all: every pick is a surface chunk
