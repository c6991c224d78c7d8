"""moleculekit_amd -- MI355X-native voxel descriptors (drop-in for one hot path of moleculekit).

    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors, getCenters
    from moleculekit_amd.occupancy_utils import calculate_occupancy
    from moleculekit_amd.batch import voxelize_lattice, voxelize_lattice_torch

Hand-written HIP kernels (gfx950) behind a C ABI (include/mkamd_voxel.h, libmkamd.so); Python is
the host side only.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.2.0"

from .voxeldescriptors import (  # noqa: F401
    getCenters, getVoxelDescriptors, rotateCoordinates, install, uninstall,
)
