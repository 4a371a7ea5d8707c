"""Extractor -- thin Python owner of one vtmc context (one per GPU / process).

Mirrors the call sequence of VoxelTerrain.BatchUpdate (VoxelTerrain.cs:365-427):
extract_* (upload + three dispatches + count read-back) then read_triangles (GetData).
numpy arrays stand in for the pinned C# arrays; nothing here computes -- every result
comes out of libvtmc.so.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import COMM_ID_BYTES, TRI_DTYPE, VERTEX_DTYPE, ChunkView, DensityParams, Modifier, VolumeBatch, VtmcError


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def elem_strides(grid):
    if grid.dtype != np.float32 or grid.ndim != 3:
        raise ValueError("grid must be a 3-D float32 array indexed [x, y, z]")
    if any(s % 4 for s in grid.strides):
        raise ValueError("grid strides must be multiples of 4 bytes")
    return tuple(int(s) // 4 for s in grid.strides)


def density_params(kind, n, seed=1337):
    """perlin3d f = 8/N; fbm8 = 8 octaves, lacunarity 2, gain 0.5, f = 4/N, minus a ramp."""
    if kind == "perlin3d":
        return DensityParams(seed, 8.0 / n, 1, 2.0, 0.5, 0.0, 0.0)
    if kind == "fbm8":
        return DensityParams(seed, 4.0 / n, 8, 2.0, 0.5, 2.0 / n, n / 2.0)
    raise ValueError("unknown density kind %r" % (kind,))


class Extractor:
    def __init__(self, device=0, lib_path=None):
        self._L = _lib.load(lib_path)   # lib_path: another build of the library beside the product's (A/B tools)
        h = ctypes.c_void_p()
        rc = self._L.vtmc_create(device, ctypes.byref(h))
        if rc != 0:
            raise VtmcError(rc, self._L.vtmc_last_error(None).decode())
        self._h = h
        self.device = device

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise VtmcError(rc, self._L.vtmc_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.vtmc_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host entry points (what the C# shim P/Invokes) --------------------------------------
    def extract_blocks(self, samples):
        """samples: (B, 1000) float32 tiles laid out as VoxelTerrain.cs:341-361.  Returns T."""
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1, 1000)
        t = ctypes.c_int32()
        self._check(self._L.vtmc_extract_blocks(self._h, _ptr(samples), samples.shape[0], ctypes.byref(t)))
        return t.value

    def extract_grid(self, grid, block_list=None):
        """grid indexed [x, y, z], shape (nx+2, ny+2, nz+2), any positive strides.  Returns T."""
        sx, sy, sz = elem_strides(grid)
        nx, ny, nz = (d - 2 for d in grid.shape)
        n = 0
        if block_list is not None:
            block_list = np.ascontiguousarray(block_list, np.int32).reshape(-1, 3)
            n = len(block_list)
        t = ctypes.c_int32()
        self._check(self._L.vtmc_extract_grid(self._h, _ptr(grid), nx, ny, nz, sx, sy, sz,
                                              _ptr(block_list), n, ctypes.byref(t)))
        return t.value

    def extract_grid_sharded(self, grid, chunk_cells, rank, world_size):
        """Returns (T_local, chunk_counts[n_local, 2] = {vertices, triangles})."""
        sx, sy, sz = elem_strides(grid)
        nx, ny, nz = (d - 2 for d in grid.shape)
        n_chunks = max(1, (nx // chunk_cells) * (ny // chunk_cells) * (nz // chunk_cells)) if chunk_cells > 0 else 1
        counts = np.zeros((n_chunks, 2), np.uint32)
        n_local, t = ctypes.c_int32(), ctypes.c_int32()
        self._check(self._L.vtmc_extract_grid_sharded(self._h, _ptr(grid), nx, ny, nz, sx, sy, sz,
                                                      chunk_cells, rank, world_size, _ptr(counts),
                                                      n_chunks, ctypes.byref(n_local), ctypes.byref(t)))
        return t.value, counts[:n_local.value].copy()

    def last_counts(self):
        b, t = ctypes.c_int32(), ctypes.c_int32()
        self._check(self._L.vtmc_last_counts(self._h, ctypes.byref(b), ctypes.byref(t)))
        return b.value, t.value

    def read_triangles(self, with_offsets=True):
        n_blocks, n_tris = self.last_counts()
        tris = np.zeros(n_tris, TRI_DTYPE)
        offs = np.zeros(n_blocks + 1, np.int32) if with_offsets else None
        self._check(self._L.vtmc_read_triangles(self._h, _ptr(tris), n_tris, _ptr(offs)))
        return (tris, offs) if with_offsets else tris

    # -- indexed (welded) output ---------------------------------------------------------------
    def set_output_mode(self, indexed):
        """indexed=True: the following extract_* / terrain_update calls produce vtmc_vertex + index
        buffers (read_indexed_mesh) instead of 76-byte records (read_triangles)."""
        self._check(self._L.vtmc_set_output_mode(self._h, 1 if indexed else 0))

    def last_vertex_count(self):
        nv = ctypes.c_int32()
        self._check(self._L.vtmc_last_vertex_count(self._h, ctypes.byref(nv)))
        return nv.value

    def read_indexed_mesh(self):
        """(vertices[V], indices[T,3] block-local, block_vertex_offsets[B+1], block_tri_offsets[B+1])."""
        n_blocks, n_tris = self.last_counts()
        nv = ctypes.c_int32()
        self._check(self._L.vtmc_last_vertex_count(self._h, ctypes.byref(nv)))
        verts = np.zeros(nv.value, VERTEX_DTYPE)
        idx = np.zeros((n_tris, 3), np.int32)
        voffs, toffs = np.zeros(n_blocks + 1, np.int32), np.zeros(n_blocks + 1, np.int32)
        self._check(self._L.vtmc_read_indexed_mesh(self._h, _ptr(verts), nv.value, _ptr(idx), n_tris, _ptr(voffs), _ptr(toffs)))
        return verts, idx, voffs, toffs

    def device_indexed_results(self):
        a, b, c, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._L.vtmc_device_indexed_results(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
        return a.value, b.value, c.value, d.value

    def read_cases(self):
        n_blocks, _ = self.last_counts()
        cases = np.zeros((n_blocks, 512), np.uint8)
        self._check(self._L.vtmc_read_cases(self._h, _ptr(cases), cases.nbytes))
        return cases

    # -- device-resident entry points ---------------------------------------------------------
    def extract_volumes_device(self, d_ptr, n, strides, n_volumes=1, volume_stride=0, stream=None, flags=0):
        """d_ptr: device address (int) of the first sample; n = (nx, ny, nz) cells per volume;
        strides = element strides (sx, sy, sz).  Returns T."""
        vb = VolumeBatch(d_ptr, n[0], n[1], n[2], strides[0], strides[1], strides[2], n_volumes, volume_stride)
        t = ctypes.c_int64()
        self._check(self._L.vtmc_extract_volumes_device(self._h, ctypes.byref(vb), stream, flags, ctypes.byref(t)))
        return t.value

    def extract_volumes_device_async(self, d_ptr, n, strides, n_volumes=1, volume_stride=0, stream=None, flags=0):
        """Queues the extract on `stream` and returns at once; extract_finish() completes it."""
        vb = VolumeBatch(d_ptr, n[0], n[1], n[2], strides[0], strides[1], strides[2], n_volumes, volume_stride)
        self._check(self._L.vtmc_extract_volumes_device_async(self._h, ctypes.byref(vb), stream, flags))

    def extract_finish(self):
        t = ctypes.c_int64()
        self._check(self._L.vtmc_extract_finish(self._h, ctypes.byref(t)))
        return t.value

    def device_results(self):
        """(triangles, block_tri_offsets, volume_counts) device addresses of the last extract."""
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._L.vtmc_device_results(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def copy_volume_counts_device(self, d_dst, capacity_volumes, stream=None):
        """Per-volume {vertices, triangles} of the last extract into a caller-owned device buffer
        (async on `stream`): what the multi-GPU driver all-gathers."""
        self._check(self._L.vtmc_copy_volume_counts_device(self._h, d_dst, capacity_volumes, stream))

    def copy_to_host(self, d_ptr, nbytes, stream=None):
        """Blocking device -> host copy through the library's own HIP runtime (vtmc_copy_to_host)."""
        out = np.empty(int(nbytes), np.uint8)
        self._check(self._L.vtmc_copy_to_host(self._h, d_ptr, _ptr(out), int(nbytes), stream))
        return out

    def copy_into_host(self, d_ptr, host_ptr, nbytes, stream=None):
        """The same copy into memory the caller owns (an address, e.g. of pinned words): nothing is allocated per call."""
        self._check(self._L.vtmc_copy_to_host(self._h, d_ptr, ctypes.c_void_p(host_ptr), int(nbytes), stream))

    def copy_u32(self, d_ptr, count, stream=None):
        return self.copy_to_host(d_ptr, 4 * int(count), stream).view(np.uint32)

    # -- multi-GPU: the RCCL all-gather of per-chunk counts behind the C ABI ---------------------
    def comm_unique_id(self):
        """128 opaque bytes drawn by rank 0 (ncclGetUniqueId); the caller distributes them."""
        buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
        rc = self._L.vtmc_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
        if rc != 0:
            raise VtmcError(rc, self._L.vtmc_last_error(None).decode())
        return bytes(buf)

    def comm_init_rank(self, unique_id, rank, world_size):
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self._L.vtmc_comm_init_rank(self._h, ctypes.cast(buf, ctypes.c_void_p), rank, world_size))

    def comm_destroy(self):
        self._check(self._L.vtmc_comm_destroy(self._h))

    def comm_share(self, owner):
        """This context issues its all-gathers through `owner`'s communicator (two contexts taking turns on one stream)."""
        self._check(self._L.vtmc_comm_share(self._h, owner._h))

    def allgather_volume_counts(self, d_all_counts, volumes_per_rank, stream=None):
        """Queues the all-gather of the last extract's per-volume {vertices, triangles} on `stream`:
        d_all_counts (device, world x volumes_per_rank x 2 u32).  Asynchronous."""
        self._check(self._L.vtmc_allgather_volume_counts(self._h, d_all_counts, volumes_per_rank, stream))

    # -- persisted chunks (device-side packer / loader) -------------------------------------------
    def chunk_write(self, path, volume, origin, with_samples=True):
        o = (ctypes.c_int32 * 3)(*[int(v) for v in origin])
        self._check(self._L.vtmc_chunk_write(self._h, str(path).encode(), volume, ctypes.byref(o), 1 if with_samples else 0))

    def chunk_read(self, path):
        """Uploads a chunk file; returns the ChunkView of device pointers (valid until the next chunk_* call)."""
        v = ChunkView()
        self._check(self._L.vtmc_chunk_read(self._h, str(path).encode(), ctypes.byref(v)))
        return v

    def reserve_triangles(self, capacity):
        self._check(self._L.vtmc_reserve_triangles(self._h, int(capacity)))

    def stream_handle(self, own_queue=True):
        """A hipStream_t of the context as an integer.  own_queue=True: the stream on a hardware queue of its own (two contexts, two steps in
        flight, steps that overlap); False: the context's own stream (what stream = None means).  Wrap it with torch.cuda.ExternalStream to
        queue torch work behind a step -- and release every torch object that touched it (events, pinned tensors copied on it: PyTorch
        records an event on the stream when it FREES such a tensor) before the context is closed: the stream dies with it."""
        h = ctypes.c_void_p()
        self._check(self._L.vtmc_context_stream(self._h, 1 if own_queue else 0, ctypes.byref(h)))
        return h.value or 0

    def last_stage_ms(self):
        ms = (ctypes.c_float * 4)()
        self._check(self._L.vtmc_last_stage_ms(self._h, ctypes.byref(ms)))
        return {"classify": ms[0], "scan": ms[1], "emit": ms[2], "total": ms[3]}

    def last_placement(self):
        """The last output-placement trial (tuning key place_outputs): ([emit ms per candidate], index kept); ([], 0) when none has run."""
        ms, n, kept = (ctypes.c_float * 16)(), ctypes.c_int32(), ctypes.c_int32()
        self._check(self._L.vtmc_last_placement(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(kept)))
        return [round(float(ms[i]), 4) for i in range(n.value)], kept.value

    def last_fill_ms(self):
        ms = ctypes.c_float()
        self._check(self._L.vtmc_last_fill_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def set_tuning(self, **kv):
        for k, v in kv.items():
            self._check(self._L.vtmc_set_tuning(self._h, k.encode(), int(v)))

    # -- device-resident terrain: VoxelTerrain.Init / Update on the GPU ------------------------
    def terrain_init(self, width, elevation, height, voxel_scale=1.0, origin=(0.0, 0.0, 0.0), seed=1):
        """VoxelTerrain.Init's grid (VoxelTerrain.cs:121-149) in HBM."""
        o = (ctypes.c_float * 3)(*origin)
        self._check(self._L.vtmc_terrain_init(self._h, width, elevation, height, voxel_scale, ctypes.byref(o), seed))
        self._terrain_dims = (width, elevation, height)

    def terrain_update(self, mods):
        """VoxelTerrain.Update (VoxelTerrain.cs:262-325) for a queue of Modifier structs.
        Returns (number of dirty blocks, T)."""
        mods = [m.to_struct() if hasattr(m, "to_struct") else m for m in mods]
        arr = (Modifier * max(len(mods), 1))()
        for i, m in enumerate(mods):
            ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(m), ctypes.sizeof(Modifier))
        nd, t = ctypes.c_int32(), ctypes.c_int32()
        self._check(self._L.vtmc_terrain_update(self._h, ctypes.cast(arr, ctypes.c_void_p), len(mods),
                                                ctypes.byref(nd), ctypes.byref(t)))
        return nd.value, t.value

    def terrain_dirty_blocks(self):
        n = ctypes.c_int32()
        self._check(self._L.vtmc_terrain_dirty_blocks(self._h, None, 0, ctypes.byref(n)))
        out = np.zeros((n.value, 3), np.int32)
        self._check(self._L.vtmc_terrain_dirty_blocks(self._h, _ptr(out), n.value, ctypes.byref(n)))
        return out

    def terrain_read_samples(self, order="x"):
        """The density grid indexed [x, y, z]; order='x': x fastest in memory, 'z': a C# float[,,]."""
        w, e, h = self._terrain_dims
        if order == "x":
            mem = np.empty((h + 2, e + 2, w + 2), np.float32)
            grid = mem.transpose(2, 1, 0)
        else:
            grid = np.empty((w + 2, e + 2, h + 2), np.float32)
        sx, sy, sz = elem_strides(grid)
        self._check(self._L.vtmc_terrain_read_samples(self._h, _ptr(grid), sx, sy, sz))
        return grid

    def density_fill_device(self, params, origins, dims, strides, volume_stride, d_out, stream=None, wait=True):
        """wait=False queues the fill on `stream` without synchronising (vtmc_density_fill_device_async)."""
        origins = np.ascontiguousarray(origins, np.int32).reshape(-1, 3)
        fn = self._L.vtmc_density_fill_device if wait else self._L.vtmc_density_fill_device_async
        self._check(fn(self._h, ctypes.byref(params), _ptr(origins), len(origins),
                                                     dims[0], dims[1], dims[2], strides[0], strides[1],
                                                     strides[2], volume_stride, d_out, stream))
