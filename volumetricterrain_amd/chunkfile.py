"""Persisted chunk format: what a streaming run leaves behind for one chunk --
its density samples (optional), its per-chunk counts and its mesh, soup or indexed.  New in the
build: the reference keeps its grid in memory only and regenerates it from the seed
(VoxelTerrain.cs:145-149, TerrainEngine.cs:56-59); a world streamed through HBM (BASELINE config 5)
needs somewhere to put its results.

Layout (little-endian, every section 16-byte aligned so it can be mapped and handed to the GPU or to
a C# reader as blittable arrays):
    header  64 B : magic "VTCHUNK1", u32 version = 1, u32 flags (1 samples, 2 soup, 4 indexed),
                   i32 origin[3] (global sample index of the chunk's first sample), i32 cells[3],
                   u32 n_blocks, u32 n_triangles, u32 n_vertices, 12 reserved bytes
    samples      : f32[(cells+2)^3], x fastest                       (flag 1)
    tri_offsets  : u32[n_blocks + 1]
    triangles    : 76-byte records (VoxelTerrain.cs:23-37)           (flag 2)
    vert_offsets : u32[n_blocks + 1], vertices 24 B, indices i32[3T] (flag 4)
"""
import struct

import numpy as np

from ._lib import TRI_DTYPE, VERTEX_DTYPE

MAGIC = b"VTCHUNK1"
HEADER = struct.Struct("<8sII3i3iIII12x")
assert HEADER.size == 64
F_SAMPLES, F_SOUP, F_INDEXED = 1, 2, 4


def _pad(f):
    f.write(b"\0" * (-f.tell() % 16))


def write_chunk(path, origin, cells, tri_offsets, samples=None, triangles=None, vertices=None, indices=None,
                vert_offsets=None):
    tri_offsets = np.ascontiguousarray(tri_offsets, np.uint32)
    n_blocks = len(tri_offsets) - 1
    flags = (F_SAMPLES if samples is not None else 0) | (F_SOUP if triangles is not None else 0) | \
            (F_INDEXED if vertices is not None else 0)
    n_tris = int(tri_offsets[-1])
    if triangles is not None and len(triangles) != n_tris:
        raise ValueError("triangles do not match tri_offsets")
    if vertices is not None and (indices is None or vert_offsets is None or len(indices) != n_tris):
        raise ValueError("indexed meshes need indices[T,3] and vert_offsets")
    with open(path, "wb") as f:
        f.write(HEADER.pack(MAGIC, 1, flags, *[int(v) for v in origin], *[int(v) for v in cells], n_blocks, n_tris,
                            0 if vertices is None else len(vertices)))
        if samples is not None:
            s = np.ascontiguousarray(samples, np.float32)
            if s.size != (cells[0] + 2) * (cells[1] + 2) * (cells[2] + 2):
                raise ValueError("samples must hold (cells+2)^3 values")
            f.write(s.tobytes())
            _pad(f)
        f.write(tri_offsets.tobytes())
        _pad(f)
        if triangles is not None:
            f.write(np.ascontiguousarray(triangles, TRI_DTYPE).tobytes())
            _pad(f)
        if vertices is not None:
            f.write(np.ascontiguousarray(vert_offsets, np.uint32).tobytes())
            _pad(f)
            f.write(np.ascontiguousarray(vertices, VERTEX_DTYPE).tobytes())
            _pad(f)
            f.write(np.ascontiguousarray(indices, np.int32).tobytes())


def read_chunk(path):
    """Returns a dict of numpy views into a read-only memory map of the file."""
    m = np.memmap(path, np.uint8, "r")
    magic, version, flags, ox, oy, oz, cx, cy, cz, n_blocks, n_tris, n_verts = HEADER.unpack(bytes(m[:64]))
    if magic != MAGIC or version != 1:
        raise ValueError("%s is not a version-1 chunk file" % path)
    out = {"origin": (ox, oy, oz), "cells": (cx, cy, cz), "flags": flags}
    pos = 64

    def take(dtype, count):
        nonlocal pos
        nbytes = np.dtype(dtype).itemsize * count
        a = m[pos:pos + nbytes].view(dtype)
        pos += nbytes + (-(pos + nbytes) % 16)
        return a

    if flags & F_SAMPLES:
        out["samples"] = take(np.float32, (cx + 2) * (cy + 2) * (cz + 2))
    out["tri_offsets"] = take(np.uint32, n_blocks + 1)
    if flags & F_SOUP:
        out["triangles"] = take(TRI_DTYPE, n_tris)
    if flags & F_INDEXED:
        out["vert_offsets"] = take(np.uint32, n_blocks + 1)
        out["vertices"] = take(VERTEX_DTYPE, n_verts)
        out["indices"] = take(np.int32, 3 * n_tris).reshape(-1, 3)
    return out
