"""Analytic density fields whose formulas are the reference's own modifiers
(TerrainModifier.cs:59-62 plane, :79-82 sphere), sampled at voxelScale = 1, origin = 0
(VoxelTerrain.cs:290: worldPos = idx * scale + origin), FP32."""
import numpy as np


def _idx(shape, order):
    nx, ny, nz = shape
    if order == "x":   # x fastest in memory, indexed [x, y, z]
        mem = np.empty((nz, ny, nx), np.float32)
        return mem.transpose(2, 1, 0)
    return np.empty((nx, ny, nz), np.float32)   # C# float[,,]: z fastest


def plane(n, h, order="x"):
    """PlaneModifier.QueryDensity: _height - pos.y"""
    g = _idx((n[0] + 2, n[1] + 2, n[2] + 2), order)
    y = np.arange(n[1] + 2, dtype=np.float32)
    g[...] = (np.float32(h) - y)[None, :, None]
    return g


def sphere(n, center, radius, order="x"):
    """SphereModifier.QueryDensity: _radius - (pos - _center).magnitude"""
    g = _idx((n[0] + 2, n[1] + 2, n[2] + 2), order)
    x = np.arange(n[0] + 2, dtype=np.float32)[:, None, None] - np.float32(center[0])
    y = np.arange(n[1] + 2, dtype=np.float32)[None, :, None] - np.float32(center[1])
    z = np.arange(n[2] + 2, dtype=np.float32)[None, None, :] - np.float32(center[2])
    mag = np.sqrt((x * x + y * y + z * z).astype(np.float32)).astype(np.float32)
    g[...] = np.float32(radius) - mag
    return g


def constant(n, value, order="x"):
    g = _idx((n[0] + 2, n[1] + 2, n[2] + 2), order)
    g[...] = np.float32(value)
    return g


def random_field(n, seed, order="x", scale=1.0):
    rng = np.random.default_rng(seed)
    g = _idx((n[0] + 2, n[1] + 2, n[2] + 2), order)
    g[...] = (rng.standard_normal(g.shape) * scale).astype(np.float32)
    return g


def all_cases_tile():
    """One 8x8x8 block whose 512 cells... cannot hold all 256 cases independently (cells share
    corners), so build 4 blocks x 64 isolated cells: cell (2i,2j,2k) of block q gets case
    q*64 + i + 4j + 16k; every other sample is -1 (outside) unless a wanted corner."""
    tiles = np.full((4, 10, 10, 10), -1.0, np.float32)   # [block, z, y, x]
    off = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    rng = np.random.default_rng(7)
    for q in range(4):
        for k in range(4):
            for j in range(4):
                for i in range(4):
                    case = q * 64 + i + 4 * j + 16 * k
                    for c, (ox, oy, oz) in enumerate(off):
                        if (case >> c) & 1:
                            tiles[q, 2 * k + oz, 2 * j + oy, 2 * i + ox] = np.float32(0.25 + rng.random())
    # negative values get distinct magnitudes so gradients never vanish
    neg = tiles < 0
    tiles[neg] = -(0.25 + rng.random(int(neg.sum()))).astype(np.float32)
    return tiles.reshape(4, 1000)
