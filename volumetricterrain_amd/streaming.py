"""Double-buffered streaming extraction over a world too large to hold at once (BASELINE config 5:
2048^3 cells of 8-octave fBm = 34 GB of samples): the chunks a rank owns are generated and
extracted batch by batch, batch k+1 being sampled on one context / stream while batch k is
extracted on the other.  New in the build -- the reference caps a world at 1025 samples per axis
(VoxelTerrain.cs:44) and has no streaming of any kind.

Two vtmc contexts, each with its own HIP stream, density buffer and result buffers; a single host
thread drives both: the fill of the next batch is queued without waiting
(vtmc_density_fill_device_async), the extract of the current batch blocks until its T is known, so the
two overlap on the device.  PyTorch only provides the device allocations.
"""
import numpy as np

from . import sharding
from .extractor import Extractor, density_params


class ChunkStream:
    def __init__(self, world_cells, chunk=128, batch_chunks=64, kind="fbm8", noise_n=None, seed=1337,
                 rank=0, world_size=1, device=0):
        import torch
        if isinstance(world_cells, int):
            world_cells = (world_cells,) * 3
        self.world, self.chunk, self.dim = tuple(world_cells), chunk, chunk + 2
        self.origins = sharding.chunk_origins(self.world, chunk, rank, world_size)
        self.batch = max(1, min(batch_chunks, len(self.origins)))
        self.params = density_params(kind, noise_n or self.world[0], seed)
        self.bpv = (chunk // 8) ** 3
        self._ex = [Extractor(device), Extractor(device)]
        with torch.cuda.device(device):
            self._buf = [torch.empty(self.batch * self.dim ** 3, dtype=torch.float32, device="cuda") for _ in range(2)]

    def close(self):
        for e in self._ex:
            e.close()
        self._buf = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def n_batches(self):
        return (len(self.origins) + self.batch - 1) // self.batch

    def _origins_of(self, k):
        return self.origins[k * self.batch:(k + 1) * self.batch]

    def _fill(self, slot, k):
        d = self.dim
        self._ex[slot].density_fill_device(self.params, self._origins_of(k), (d, d, d), (1, d, d * d), d ** 3,
                                           self._buf[slot].data_ptr(), None, wait=False)

    def batches(self):
        """Yields (k, origins, T, extractor): the extractor still holds batch k's results (triangles,
        block offsets, per-chunk counts) until the generator is advanced twice more."""
        nb = self.n_batches()
        if nb == 0:
            return
        d, c = self.dim, self.chunk
        self._fill(0, 0)
        for k in range(nb):
            slot = k & 1
            if k + 1 < nb:
                self._fill(slot ^ 1, k + 1)   # sampled while batch k is extracted below
            org = self._origins_of(k)
            T = self._ex[slot].extract_volumes_device(self._buf[slot].data_ptr(), (c, c, c), (1, d, d * d), len(org), d ** 3)
            yield k, org, T, self._ex[slot]

    def run(self):
        """Drains the stream; returns (total triangles, per-chunk {vertices, triangles} in owned-chunk order)."""
        counts, total = [], 0
        for _, org, T, ex in self.batches():
            _, _, vc_ptr = ex.device_results()
            counts.append(ex.copy_u32(vc_ptr, 2 * len(org)).reshape(-1, 2).astype(np.int64))
            total += T
        return total, (np.concatenate(counts) if counts else np.zeros((0, 2), np.int64))
