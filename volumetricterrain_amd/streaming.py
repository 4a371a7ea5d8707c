"""Double-buffered streaming extraction over a world too large to hold at once (BASELINE config 5:
2048^3 cells of 8-octave fBm = 34 GB of samples): the chunks a rank owns are generated and
extracted batch by batch.  New in the build -- the reference caps a world at 1025 samples per axis
(VoxelTerrain.cs:44) and has no streaming of any kind.

Two vtmc contexts (density buffer + result buffers each) fed by a single host thread that stays a batch
ahead: sample(k + 1) and extract(k) are queued (vtmc_density_fill_device_async,
vtmc_extract_volumes_device_async) before the host takes batch k - 1's result (vtmc_extract_finish waits
for that extract's own event only), so the device goes from kernel to kernel without waiting for the host.
Each context queues its sampler and its extract on its own-queue stream (round 5: a hardware queue each;
rounds 2-4 used one stream for both -- two ordinary streams shared a queue and ran in turn, or lost 4 % with
the sampler's residency cut to make room): the batches' kernels overlap where one drains and the next ramps
up, 18.9 against 19.8 ms per 2048^3 pass.  PyTorch only provides the allocations.
"""
import numpy as np

from . import sharding
from .extractor import Extractor, density_params


class ChunkStream:
    def __init__(self, world_cells, chunk=128, batch_chunks=64, kind="fbm8", noise_n=None, seed=1337,
                 rank=0, world_size=1, device=0, sampler_wgs_per_cu=None, two_queues=True):
        import torch
        if isinstance(world_cells, int):
            world_cells = (world_cells,) * 3
        self.world, self.chunk, self.dim = tuple(world_cells), chunk, chunk + 2
        self.origins = sharding.chunk_origins(self.world, chunk, rank, world_size)
        self.batch = max(1, min(batch_chunks, len(self.origins)))
        self.params = density_params(kind, noise_n or self.world[0], seed)
        self.bpv = (chunk // 8) ** 3
        self._ex = [Extractor(device), Extractor(device)]
        for e in self._ex:   # the sampler leaves the samples' sign bits; the classify stage of the same (unmodified) buffer reads those
            e.set_tuning(fill_keeps_signs=1)
        if sampler_wgs_per_cu is not None:   # residency of the (ALU-bound) sampler: what it leaves free, the other stream's extract uses
            for e in self._ex:
                e.set_tuning(density_wgs_per_cu=int(sampler_wgs_per_cu))
        with torch.cuda.device(device):
            self._buf = [torch.empty(self.batch * self.dim ** 3, dtype=torch.float32, device="cuda") for _ in range(2)]
            self._stream = None if two_queues else torch.cuda.Stream()   # rounds 2-4: one torch stream for both contexts
        # two_queues (default): each context's sampler and extract on the context's own-queue stream (a hardware queue each,
        # vtmc_context_stream) instead of one stream for both: S0 E0 S2 E2 ... beside S1 E1 S3 E3 ...; 18.9 against 19.8 ms per 2048^3 pass
        # (profiles/r05/stream2048_two_queues.txt)
        self._sptr = [e.stream_handle() for e in self._ex] if two_queues else [self._stream.cuda_stream] * 2
        with torch.cuda.device(device):
            self._copy_stream = torch.cuda.Stream() if two_queues else None   # the host's read-backs never queue behind the next batch's sampler

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
                                           self._buf[slot].data_ptr(), self._sptr[slot], wait=False)

    def batches(self):
        """Yields (k, origins, T, extractor): the extractor holds batch k's results (triangles, block offsets,
        per-chunk counts) ONLY UNTIL THE GENERATOR IS NEXT ADVANCED -- batch k + 2 is extracted by the same context,
        and the very next advance queues it over the same result buffers.  A consumer copies what it keeps before it
        asks for the next batch (run() does); device pointers from device_results() must not be held across an advance.
        Stream order: S0 S1 E0 E1 S2 E2 S3 E3 ... (S = sample, E = extract); the host takes E(k - 1) while E(k) runs."""
        nb = self.n_batches()
        if nb == 0:
            return
        d, c = self.dim, self.chunk
        self._fill(0, 0)
        if nb > 1:
            self._fill(1, 1)
        for k in range(nb + 1):
            if k < nb:
                self._ex[k & 1].extract_volumes_device_async(self._buf[k & 1].data_ptr(), (c, c, c), (1, d, d * d),
                                                             len(self._origins_of(k)), d ** 3, self._sptr[k & 1])
            if k >= 1:
                ex = self._ex[(k - 1) & 1]
                T = ex.extract_finish()                 # batch k - 1 is done; batch k's extract is already queued behind it
                if k + 1 < nb:
                    self._fill((k + 1) & 1, k + 1)      # its buffer is free again
                yield k - 1, self._origins_of(k - 1), T, ex

    def run(self):
        """Drains the stream; returns (total triangles, per-chunk {vertices, triangles} in owned-chunk order)."""
        counts, total = [], 0
        for _, org, T, ex in self.batches():
            _, _, vc_ptr = ex.device_results()
            counts.append(ex.copy_u32(vc_ptr, 2 * len(org), self._copy_stream.cuda_stream if self._copy_stream is not None else None).reshape(-1, 2).astype(np.int64))
            total += T
        return total, (np.concatenate(counts) if counts else np.zeros((0, 2), np.int64))
