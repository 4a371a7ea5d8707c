"""Host-side mirror of the reference's TerrainModifier classes (TerrainModifier.cs:19-152): same
names, fields and bound formulas, flattened into the vtmc_modifier struct the C ABI takes.

Only what crosses the boundary lives here -- the AABB properties (LowerBound / UpperBound, which
the reference evaluates on the host, VoxelTerrain.cs:273-279) and the parameters of QueryDensity;
the density itself is evaluated on the GPU (csrc/terrain.hip).  All arithmetic is FP32, in the
order of the C# expressions.
"""
import numpy as np

from ._lib import MOD_CYLINDER, MOD_HEIGHTMAP, MOD_PLANE, MOD_SPHERE, Modifier

_f = np.float32
FLOAT_MIN_VALUE = _f(-3.4028234663852886e38)  # C# float.MinValue


def _vec(v):
    return np.asarray(v, _f).reshape(3)


def _dot(a, b):
    """UnityEngine.Vector3.Dot: a.x*b.x + a.y*b.y + a.z*b.z, FP32, left to right."""
    return _f(_f(_f(a[0] * b[0]) + _f(a[1] * b[1])) + _f(a[2] * b[2]))


def _project_on_plane(v, n):
    """UnityEngine.Vector3.ProjectOnPlane: v - n * Dot(v, n) / Dot(n, n)."""
    return (v - n * (_dot(v, n) / _dot(n, n))).astype(_f)


class TerrainModifier:
    AddOrErode = True  # true -> add (union), false -> erode (difference): TerrainModifier.cs:31-32

    def to_struct(self):
        m = Modifier(self.kind, 1 if self.AddOrErode else 0)
        m.lower[:] = tuple(float(x) for x in self.LowerBound)
        m.upper[:] = tuple(float(x) for x in self.UpperBound)
        p = self.params()
        m.p[0:len(p)] = tuple(float(x) for x in p)
        self.attach(m)
        return m

    def attach(self, m):
        """Hook for modifiers that carry an array (the heightmap)."""


class PlaneModifier(TerrainModifier):
    """f(x,y,z) = y0 - y (TerrainModifier.cs:38-65)."""
    kind = MOD_PLANE

    def __init__(self, height, low, up, addOrErode=True):
        if low[0] > up[0] or low[1] > up[1]:
            raise ValueError("invalud aabb")  # sic, TerrainModifier.cs:52
        self._height, self._low, self._up = _f(height), np.asarray(low, _f), np.asarray(up, _f)
        self.AddOrErode = addOrErode

    @property
    def LowerBound(self):
        return np.array([self._low[0], FLOAT_MIN_VALUE, self._low[1]], _f)

    @property
    def UpperBound(self):
        return np.array([self._up[0], self._height + _f(1), self._up[1]], _f)

    def params(self):
        return [self._height]


class SphereModifier(TerrainModifier):
    """f = r - |p - c| (TerrainModifier.cs:70-91)."""
    kind = MOD_SPHERE

    def __init__(self, center, radius, addOrErode=True):
        self._center, self._radius = _vec(center), _f(radius)
        self.AddOrErode = addOrErode

    @property
    def LowerBound(self):
        return (self._center - self._radius).astype(_f)

    @property
    def UpperBound(self):
        return (self._center + self._radius).astype(_f)

    def params(self):
        return [self._center[0], self._center[1], self._center[2], self._radius]


class CylinderModifier(TerrainModifier):
    """Capped cylinder along _axisDir (TerrainModifier.cs:96-152)."""
    kind = MOD_CYLINDER

    def __init__(self, start, direction, length, radius, addOrErode=True):
        d = _vec(direction)
        self._axisStart = _vec(start)
        self._axisDir = (d / _f(np.sqrt(_dot(d, d)))).astype(_f)  # dir.normalized
        self._axisLength, self._radius = _f(length), _f(radius)
        self.AddOrErode = addOrErode

    def _bound(self, sign):
        end = (self._axisStart + self._axisDir * self._axisLength).astype(_f)
        out = np.zeros(3, _f)
        for a in range(3):
            unit = np.zeros(3, _f)
            unit[a] = sign
            shift = _project_on_plane(unit, self._axisDir) * self._radius
            # LowerBound: dir > 0 ? start : end;  UpperBound: dir < 0 ? start : end (TerrainModifier.cs:104-131)
            from_start = self._axisDir[a] > 0 if sign < 0 else self._axisDir[a] < 0
            out[a] = ((self._axisStart if from_start else end) + shift)[a]
        return out

    @property
    def LowerBound(self):
        return self._bound(-1.0)

    @property
    def UpperBound(self):
        return self._bound(1.0)

    def params(self):
        return [*self._axisStart, *self._axisDir, self._axisLength, self._radius]


class IslandModifier(TerrainModifier):
    """The heightmap modifier of the world build (IslandModifier.cs:34-92, inserted at
    TerrainEngine.cs:87): density = bilinear(_heightmap)(x, z) - y.  The reference fills _heightmap
    from Island.GetElevation (island generation: out of scope here), so this mirror takes the
    float[widthRes, heightRes] array itself plus _island.width / .height / ._maxElevation."""
    kind = MOD_HEIGHTMAP

    def __init__(self, heightmap, island_width, island_height, max_elevation, addOrErode=True):
        self._heightmap = np.ascontiguousarray(heightmap, _f)
        if self._heightmap.ndim != 2:
            raise ValueError("heightmap must be a 2-D array indexed [u, v]")
        self._width, self._height, self._maxElevation = _f(island_width), _f(island_height), _f(max_elevation)
        self.AddOrErode = addOrErode

    @property
    def LowerBound(self):
        return np.array([0, FLOAT_MIN_VALUE, 0], _f)

    @property
    def UpperBound(self):
        return np.array([self._width, self._maxElevation, self._height], _f)

    def params(self):
        return [self._width, self._height]

    def attach(self, m):
        m.data = self._heightmap.ctypes.data   # borrowed: this object outlives the call
        m.data_dims[:] = self._heightmap.shape
