"""ctypes binding of libvtmc.so -- the same C ABI a C# host binds with [DllImport].

There is no CPU fallback: if the library is missing it is built with hipcc; if that fails, or the
library cannot be loaded, importing raises.  Creating a context without a HIP device fails with
VtmcError (vtmc_create returns VTMC_ERR_DEVICE).
"""
import ctypes
import os
import sys

import numpy as np

from . import build as _build

TRI_DTYPE = np.dtype([("p0", "<f4", 3), ("p1", "<f4", 3), ("p2", "<f4", 3),
                      ("n0", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3),
                      ("block", "<i4")])
assert TRI_DTYPE.itemsize == 76  # CSTriangle.stride, VoxelTerrain.cs:36

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3)])   # vtmc_vertex
assert VERTEX_DTYPE.itemsize == 24
OUTPUT_SOUP, OUTPUT_INDEXED = 0, 1

OK = 0
ERR_INVALID_ARG, ERR_DIMS, ERR_CAPACITY, ERR_DEVICE, ERR_NO_RESULT, ERR_TOO_LARGE = -1, -2, -3, -4, -5, -6
FLAG_WANT_CASES, FLAG_NO_DENSE_PATH = 1, 2

# every symbol include/vtmc.h declares (tests/test_abi_symbols.py checks the header against this)
SYMBOLS = [
    "vtmc_version", "vtmc_create", "vtmc_destroy", "vtmc_last_error", "vtmc_extract_blocks",
    "vtmc_extract_grid", "vtmc_extract_grid_sharded", "vtmc_read_triangles", "vtmc_read_cases",
    "vtmc_last_counts", "vtmc_extract_volumes_device", "vtmc_device_results",
    "vtmc_reserve_triangles", "vtmc_last_stage_ms", "vtmc_set_tuning", "vtmc_density_fill_device",
    "vtmc_terrain_init", "vtmc_terrain_update", "vtmc_terrain_dirty_blocks", "vtmc_terrain_read_samples",
    "vtmc_terrain_device_grid", "vtmc_copy_volume_counts_device", "vtmc_density_fill_device_async",
    "vtmc_set_output_mode", "vtmc_last_vertex_count", "vtmc_read_indexed_mesh", "vtmc_device_indexed_results",
    "vtmc_comm_unique_id", "vtmc_comm_init_rank", "vtmc_comm_destroy", "vtmc_comm_share", "vtmc_allgather_volume_counts",
    "vtmc_copy_to_host", "vtmc_chunk_write", "vtmc_chunk_read",
    "vtmc_extract_volumes_device_async", "vtmc_extract_finish", "vtmc_last_fill_ms", "vtmc_context_stream", "vtmc_release_streams", "vtmc_last_placement",
]
COMM_ID_BYTES = 128

MOD_PLANE, MOD_SPHERE, MOD_CYLINDER, MOD_HEIGHTMAP = 0, 1, 2, 3


class Modifier(ctypes.Structure):
    """vtmc_modifier: one queued TerrainModifier (TerrainModifier.cs:19-33), bounds as the C#
    LowerBound / UpperBound properties return them."""
    _fields_ = [("kind", ctypes.c_int32), ("add_or_erode", ctypes.c_int32), ("lower", ctypes.c_float * 3),
                ("upper", ctypes.c_float * 3), ("p", ctypes.c_float * 8), ("data", ctypes.c_void_p),
                ("data_dims", ctypes.c_int32 * 2)]


class VolumeBatch(ctypes.Structure):
    _fields_ = [("d_samples", ctypes.c_void_p), ("nx", ctypes.c_int32), ("ny", ctypes.c_int32),
                ("nz", ctypes.c_int32), ("stride_x", ctypes.c_int64), ("stride_y", ctypes.c_int64),
                ("stride_z", ctypes.c_int64), ("n_volumes", ctypes.c_int32),
                ("volume_stride", ctypes.c_int64)]


class ChunkView(ctypes.Structure):
    """vtmc_chunk_view: device pointers into an uploaded chunk-file image."""
    _fields_ = [("origin", ctypes.c_int32 * 3), ("cells", ctypes.c_int32 * 3), ("flags", ctypes.c_uint32),
                ("n_blocks", ctypes.c_uint32), ("n_triangles", ctypes.c_uint32), ("n_vertices", ctypes.c_uint32),
                ("d_samples", ctypes.c_void_p), ("d_tri_offsets", ctypes.c_void_p), ("d_triangles", ctypes.c_void_p),
                ("d_vert_offsets", ctypes.c_void_p), ("d_vertices", ctypes.c_void_p), ("d_indices", ctypes.c_void_p)]


class DensityParams(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("frequency", ctypes.c_float), ("octaves", ctypes.c_int32),
                ("lacunarity", ctypes.c_float), ("gain", ctypes.c_float),
                ("ramp_scale", ctypes.c_float), ("ramp_center", ctypes.c_float)]


class VtmcError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("vtmc error %d: %s" % (code, text))
        self.code = code


_lib = None


def load(path=None):
    """Load (building if stale) libvtmc.so and declare the prototypes.  `path`: another build of the library, loaded beside the product's
    and not cached (tools/ab_two_libs.py: two builds alternating in ONE process -- boxes of the pool drift by several percent between
    processes)."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    explicit = path is not None
    path = path or os.environ.get("VTMC_LIB") or _build.build()   # VTMC_LIB: A/B of two builds on one box (tools/ab_bench.py)
    # PyTorch wheels bundle their own libamdhip64; if this process is going to use torch as well
    # (device memory, streams, torch.distributed), torch must load first so both bind to ONE HIP
    # runtime -- loaded the other way round torch.cuda reports no device.
    if "torch" not in sys.modules and os.environ.get("VTMC_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(path)
    vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
    P = ctypes.POINTER
    L.vtmc_version.restype = ctypes.c_char_p
    L.vtmc_create.argtypes = [i32, P(vp)]
    L.vtmc_destroy.argtypes = [vp]
    L.vtmc_last_error.argtypes = [vp]
    L.vtmc_last_error.restype = ctypes.c_char_p
    L.vtmc_extract_blocks.argtypes = [vp, vp, i32, P(i32)]
    L.vtmc_extract_grid.argtypes = [vp, vp, i32, i32, i32, i64, i64, i64, vp, i32, P(i32)]
    L.vtmc_extract_grid_sharded.argtypes = [vp, vp, i32, i32, i32, i64, i64, i64, i32, i32, i32,
                                            vp, i32, P(i32), P(i32)]
    L.vtmc_read_triangles.argtypes = [vp, vp, i64, vp]
    L.vtmc_read_cases.argtypes = [vp, vp, i64]
    L.vtmc_last_counts.argtypes = [vp, P(i32), P(i32)]
    L.vtmc_extract_volumes_device.argtypes = [vp, P(VolumeBatch), vp, u32, P(i64)]
    L.vtmc_extract_volumes_device_async.argtypes = [vp, P(VolumeBatch), vp, u32]
    L.vtmc_extract_finish.argtypes = [vp, P(i64)]
    L.vtmc_last_fill_ms.argtypes = [vp, P(ctypes.c_float)]
    if not explicit or hasattr(L, "vtmc_context_stream"):   # an older build loaded beside the product's (A/B tools) may lack the newest entry points
        L.vtmc_context_stream.argtypes = [vp, i32, P(vp)]
    if not explicit or hasattr(L, "vtmc_release_streams"):
        L.vtmc_release_streams.argtypes = []
    if not explicit or hasattr(L, "vtmc_last_placement"):
        L.vtmc_last_placement.argtypes = [vp, P(ctypes.c_float * 16), P(i32), P(i32)]
    L.vtmc_device_results.argtypes = [vp, P(vp), P(vp), P(vp)]
    L.vtmc_reserve_triangles.argtypes = [vp, i64]
    L.vtmc_copy_volume_counts_device.argtypes = [vp, vp, i32, vp]
    L.vtmc_last_stage_ms.argtypes = [vp, P(ctypes.c_float * 4)]
    L.vtmc_set_tuning.argtypes = [vp, ctypes.c_char_p, i32]
    L.vtmc_density_fill_device.argtypes = [vp, P(DensityParams), vp, i32, i32, i32, i32,
                                           i64, i64, i64, i64, vp, vp]
    L.vtmc_density_fill_device_async.argtypes = L.vtmc_density_fill_device.argtypes
    L.vtmc_set_output_mode.argtypes = [vp, i32]
    L.vtmc_last_vertex_count.argtypes = [vp, P(i32)]
    L.vtmc_read_indexed_mesh.argtypes = [vp, vp, i64, vp, i64, vp, vp]
    L.vtmc_device_indexed_results.argtypes = [vp, P(vp), P(vp), P(vp), P(vp)]
    L.vtmc_terrain_init.argtypes = [vp, i32, i32, i32, ctypes.c_float, P(ctypes.c_float * 3), ctypes.c_uint64]
    L.vtmc_terrain_update.argtypes = [vp, vp, i32, P(i32), P(i32)]
    L.vtmc_terrain_dirty_blocks.argtypes = [vp, vp, i32, P(i32)]
    L.vtmc_terrain_read_samples.argtypes = [vp, vp, i64, i64, i64]
    L.vtmc_terrain_device_grid.argtypes = [vp, P(vp), P(i64 * 3), P(i32 * 3)]
    L.vtmc_comm_unique_id.argtypes = [vp]
    L.vtmc_comm_init_rank.argtypes = [vp, vp, i32, i32]
    L.vtmc_comm_destroy.argtypes = [vp]
    L.vtmc_comm_share.argtypes = [vp, vp]
    L.vtmc_allgather_volume_counts.argtypes = [vp, vp, i32, vp]
    L.vtmc_copy_to_host.argtypes = [vp, vp, vp, i64, vp]
    L.vtmc_chunk_write.argtypes = [vp, ctypes.c_char_p, i32, P(i32 * 3), i32]
    L.vtmc_chunk_read.argtypes = [vp, ctypes.c_char_p, P(ChunkView)]
    for name in SYMBOLS:
        if explicit and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int:
            fn.restype = i32
    if not explicit:
        _lib = L
    return L


def release_streams():
    """vtmc_release_streams: destroys the streams the library keeps parked for contexts that are gone.  Call it once nothing of the host's
    (events, stream wrappers, pinned tensors copied on them) refers to a handle of a closed Extractor any more -- the tools that run under
    rocprofv3 do, before they exit."""
    return _lib.vtmc_release_streams() if _lib is not None else 0


def library_path():
    return _build.LIB if os.path.exists(_build.LIB) else None
