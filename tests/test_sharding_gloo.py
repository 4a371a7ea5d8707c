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
