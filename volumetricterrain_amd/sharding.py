"""Chunk sharding across the GPUs of a node -- new in the build; the reference is
single-process, single-GPU (no collective call site exists in it).

Unit = one chunk of chunk_cells^3 cells with its +2 sample halo (independent: a block only needs
its own 10^3 samples, VoxelTerrain.cs:346-359).  Chunk c belongs to rank c % world_size.  The only
exchange is one all-gather of the per-chunk {vertex count, triangle count} pairs, after which every
rank derives the global offsets with a local exclusive scan.  No density or mesh data crosses GPUs.
"""
import numpy as np


def chunk_grid(n, chunk_cells):
    if isinstance(n, int):
        n = (n, n, n)
    if any(d % chunk_cells for d in n):
        raise ValueError("chunk size %d must divide the grid %r" % (chunk_cells, n))
    return tuple(d // chunk_cells for d in n)


def chunk_origins(n, chunk_cells, rank=0, world_size=1):
    """Global sample origin of every chunk owned by `rank`: chunk c = cx + ncx*(cy + ncy*cz)."""
    ncx, ncy, ncz = chunk_grid(n, chunk_cells)
    out = []
    for c in range(ncx * ncy * ncz):
        if c % world_size != rank:
            continue
        cx, cy, cz = c % ncx, (c // ncx) % ncy, c // (ncx * ncy)
        out.append((cx * chunk_cells, cy * chunk_cells, cz * chunk_cells))
    return np.asarray(out, np.int32).reshape(-1, 3)


def owned_chunks(n_chunks, rank, world_size):
    return list(range(rank, n_chunks, world_size))


def global_offsets(all_counts):
    """all_counts: (n_chunks, 2) {vertices, triangles} in global chunk order -> exclusive scans."""
    c = np.asarray(all_counts, np.int64).reshape(-1, 2)
    off = np.zeros((len(c) + 1, 2), np.int64)
    np.cumsum(c, axis=0, out=off[1:])
    return off


def interleave_rank_counts(per_rank_counts, world_size):
    """Re-order gathered per-rank arrays (rank r holds chunks r, r+W, r+2W, ...) into chunk order."""
    n = sum(len(p) for p in per_rank_counts)
    out = np.zeros((n, 2), np.int64)
    for r, p in enumerate(per_rank_counts):
        out[r::world_size][:len(p)] = np.asarray(p).reshape(-1, 2)
    return out


def allgather_counts(local_counts, group=None):
    """All-gather of per-chunk counts over torch.distributed (backend 'nccl' = RCCL over xGMI on the
    GPU box, 'gloo' in CPU tests).  local_counts: torch tensor (n_local, 2), same n_local on every
    rank (pad with zeros otherwise).  Returns the (world, n_local, 2) gathered tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = local_counts.contiguous().view(-1)
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)   # one collective; flat views suit nccl and gloo alike
    return out.view((world,) + tuple(local_counts.shape))
