"""volumetricterrain_amd -- MI355X-native marching-cubes extraction path of
MangoSister/VolumetricTerrain (VoxelTerrain.BatchUpdate + its three compute kernels) as
hand-written HIP for gfx950 behind the C ABI of include/vtmc.h.

Only the hot path lives here: csrc/ (HIP kernels + C ABI), the ctypes binding,
the host-side mirror of the reference's VoxelTerrain chunk API, and chunk sharding helpers.
"""
from ._lib import TRI_DTYPE, VERTEX_DTYPE, VtmcError, load, library_path, release_streams  # noqa: F401
from .extractor import Extractor, density_params, elem_strides  # noqa: F401
from .modifiers import CylinderModifier, IslandModifier, PlaneModifier, SphereModifier  # noqa: F401

__all__ = ["CylinderModifier", "IslandModifier", "PlaneModifier", "SphereModifier", "Extractor", "TRI_DTYPE", "VERTEX_DTYPE", "VtmcError", "density_params", "elem_strides", "load", "library_path", "release_streams"]
