"""Drop-in for the voxel-descriptor path of ``moleculekit.tools.voxeldescriptors`` on MI355X.

Same function names, arguments, error behaviour and returned ndarray layout as the reference
(tools/voxeldescriptors.py): ``getVoxelDescriptors`` (:251-365), ``getCenters`` (:197-248),
``rotateCoordinates`` (:78-114), ``_getGridCenters`` (:125-132), ``_getChannelRadii`` (:117-121),
``_getOccupancyC`` (:515-533).  The occupancy computation itself runs in hand-written HIP kernels
behind ``libmkamd.so`` (``method="C"`` -- the reference's only accepted value -- and ``method="HIP"``
both select them and fail loudly without a GPU; ``method="CPU"`` explicitly selects the library's host implementation,
``mkamd_calculate_occupancy_cpu``: double precision, bit-identical to the reference, for hosts without a device).

Return order is the reference's: ``(features, centers)`` with ``usercenters`` else
``(features, centers, nvoxels)``; features float64 ``[V, C]`` (float32-accurate values, <= 1e-5
from the reference's float64), V flattened x slowest / z fastest, channel order ``_order``.

Channels: pass ``userchannels`` (bool masks or float sigmas); without them the
table-driven typing of ``moleculekit_amd.channels`` is used (molecules that already carry PDBQT atom types) --
always, whatever else is installed; ``getChannels(..., backend="moleculekit")`` / ``CHANNELS_BACKEND`` delegate to
an installed moleculekit explicitly.  Assigning atom types (OpenBabel) and SmallMol typing (RDKit) are outside
this package.
"""
from __future__ import annotations

import ctypes
import logging
from functools import lru_cache

import numpy as np

from . import batch as _batch
from .util import boundingBox, rotationMatrix

logger = logging.getLogger(__name__)

_order = (
    "hydrophobic", "aromatic", "hbond_acceptor", "hbond_donor",
    "positive_ionizable", "negative_ionizable", "metal", "occupancies",
)


def rotateCoordinates(coords, rotations, center):
    """Rotate ``coords`` (natoms, 3) about ``center`` by the angles ``rotations=[rx, ry, rz]``
    (radians), applied one after the other around x, then y, then z (voxeldescriptors.py:78-114)."""
    angles = list(rotations)
    out = np.array(coords, copy=True)
    for axis, ang in zip(([1, 0, 0], [0, 1, 0], [0, 0, 1]), angles):
        rot = rotationMatrix(axis, ang)
        out = np.dot(out - center, np.transpose(rot)) + center
    return out


def _getChannelRadii(molelements):
    """Per-atom van der Waals radius from the element symbols (voxeldescriptors.py:117-121)."""
    from ._vdw_radii import VDW_RADIUS

    return np.array([VDW_RADIUS[e] for e in molelements])


@lru_cache(maxsize=10)
def _getGridCenters(x, y, z, resolution):
    """Lattice offsets ``index * resolution`` as float64 [x, y, z, 3] (voxeldescriptors.py:125-132);
    z is the fastest axis.  Cached like the reference (same grids recur for every molecule)."""
    out = np.empty((int(x), int(y), int(z), 3), dtype=np.float64)
    out[..., 0] = (np.arange(x) * resolution)[:, None, None]
    out[..., 1] = (np.arange(y) * resolution)[None, :, None]
    out[..., 2] = (np.arange(z) * resolution)[None, None, :]
    return out


def _gridSpec(mol=None, buffer=0, boxsize=None, center=None, voxelsize=1):
    """bb_min and nvoxels of the grid ``getCenters`` would build (voxeldescriptors.py:232-243).

    bbox branch: float32 min/max of the coordinates, expanded by ``buffer`` IN float32, one extra
    voxel (`+1`); boxsize branch: ``ceil(boxsize/voxelsize)`` voxels starting at center-boxsize/2.
    Voxel 0's centre sits AT bb_min (no half-voxel shift)."""
    if boxsize is None:
        bb_min, bb_max = boundingBox(mol)
        bb_min -= buffer
        bb_max += buffer
        nvoxels = np.ceil((bb_max - bb_min) / voxelsize).astype(int) + 1
    else:
        boxsize = np.array(boxsize)
        center = np.array(center)
        nvoxels = np.ceil(boxsize / voxelsize).astype(int)
        bb_min = center - (boxsize / 2)
    return bb_min, nvoxels


_CENTERS_LOCK = __import__("threading").Lock()
_CENTERS_CACHE = {}          # (bb_min bytes, nvoxels, voxelsize) -> centres; a handful of entries (one pocket, many poses)


def _centersFromSpec(bb_min, nvoxels, voxelsize):
    """``(lattice + bb_min).reshape(V, 3).copy()`` of the reference (:245-247) in one pass over the array: the sum
    is written straight into the [V, 3] result (same values, float64, C-contiguous, owns its data).  Repeated calls on
    the same grid (a screening loop voxelizes thousands of poses in one pocket box) get a copy of the cached array
    -- a memcpy instead of the broadcast add; every caller still owns what it gets, like with the reference."""
    nx, ny, nz = (int(v) for v in nvoxels)
    bb = np.asarray(bb_min)
    key = (bb.dtype.str, bb.tobytes(), nx, ny, nz, float(voxelsize))
    with _CENTERS_LOCK:                         # the drop-in API is per-thread safe: the cache is shared, so guarded
        hit = _CENTERS_CACHE.get(key)
    if hit is not None:
        return hit.copy()
    centers = np.empty((nx * ny * nz, 3), dtype=np.float64)
    np.add(_getGridCenters(nx, ny, nz, voxelsize), bb_min, out=centers.reshape(nx, ny, nz, 3))
    if centers.nbytes <= (64 << 20):
        keep = centers.copy()
        with _CENTERS_LOCK:
            while len(_CENTERS_CACHE) >= 4:
                _CENTERS_CACHE.pop(next(iter(_CENTERS_CACHE)), None)
            _CENTERS_CACHE[key] = keep
    return centers


def getCenters(mol=None, buffer=0, boxsize=None, center=None, voxelsize=1):
    """Voxel centres for voxelization: (centers float64 [V,3], nvoxels int (3,)) -- same arguments
    and results as the reference (voxeldescriptors.py:197-248)."""
    bb_min, nvoxels = _gridSpec(mol, buffer, boxsize, center, voxelsize)
    return _centersFromSpec(bb_min, nvoxels, voxelsize), nvoxels


CHANNELS_BACKEND = "table"   # what getVoxelDescriptors(mol) without `userchannels` uses; see getChannels


def getChannels(mol, aromaticNitrogen=False, version=2, validitychecks=True, backend=None):
    """Property channels of a molecule (voxeldescriptors.py:135-194).

    ``backend="table"`` (the default, ``CHANNELS_BACKEND``): this package's table-driven typing
    (``moleculekit_amd.channels``) for molecules that already carry PDBQT atom types -- the same code whatever
    is installed next to it.  ``backend="moleculekit"``: delegate, explicitly, to an installed moleculekit
    (which can re-type with OpenBabel / RDKit); raises ImportError when it is not importable."""
    backend = CHANNELS_BACKEND if backend is None else backend
    if backend == "table":
        from .channels import getChannels as table_getChannels

        return table_getChannels(mol, aromaticNitrogen, version, validitychecks)
    if backend == "moleculekit":
        from moleculekit.tools.voxeldescriptors import getChannels as ref_getChannels

        return ref_getChannels(mol, aromaticNitrogen, version, validitychecks)
    raise ValueError(f"unknown channels backend {backend!r} (expected 'table' or 'moleculekit')")


def getVoxelDescriptors(mol, boxsize=None, voxelsize=1, buffer=0, center=None, usercenters=None,
                        userchannels=None, usercoords=None, aromaticNitrogen=False, method="C",
                        version=2, validitychecks=True):
    """Voxel descriptors of a molecule -- the reference's signature and semantics
    (voxeldescriptors.py:251-365), computed on the MI355X.

    Returns ``(features, centers)`` when ``usercenters`` is given, else
    ``(features, centers, nvoxels)``.
    """
    channels = userchannels
    if channels is None:
        channels, mol = getChannels(mol, aromaticNitrogen, version, validitychecks)
    channels = np.asarray(channels)

    if channels.dtype == bool:   # bool masks -> per-channel sigma = vdW radius of the atom (:332-335)
        sigmas = _getChannelRadii(mol.element)
        channels = sigmas[:, np.newaxis] * channels.astype(float)

    nvoxels = None
    lattice = None
    centers = usercenters
    if centers is None:
        bb_min, nvoxels = _gridSpec(mol, buffer, boxsize, center, voxelsize)
        lattice = (np.asarray(bb_min, dtype=np.float64), nvoxels, voxelsize)     # (the centres themselves: below, beside the device)

    coords = usercoords
    if coords is None:
        coords = mol.get("coords")
    coords = np.asarray(coords)
    if coords.ndim == 3:
        if coords.shape[2] != 1:
            raise RuntimeError(
                "Only a single set of coordinates should be passed for voxelixation. "
                "Make sure your coordinates are either 3D with a last dim of 1 or 2D.")
        coords = coords[:, :, 0]

    if method.upper() not in ("C", "HIP", "CPU"):
        raise RuntimeError("As of moleculekit 0.9.2 we only support C implementation of voxelization")
    if method.upper() == "CPU":
        # the library's host implementation (explicit only: "C" and "HIP" never fall back to it) -- the reference's own
        # _getOccupancyC steps (:515-533) around mkamd_calculate_occupancy_cpu
        if lattice is not None:
            centers = _centersFromSpec(bb_min, nvoxels, voxelsize)
        features = _getOccupancyCPU(coords, centers, channels)
    elif lattice is not None:
        # the grid is ours: the kernels are enqueued first, the centres (a copy of the cached array: 330 KB for a 24^3
        # grid) are made while the device computes, then the features are taken out
        finish = _occupancyLatticeBegin(coords, channels, lattice)
        try:
            centers = _centersFromSpec(bb_min, nvoxels, voxelsize)
        finally:
            features = finish()
    else:
        features = _getOccupancyC(coords, centers, channels)

    if nvoxels is None:
        return features, centers
    return features, centers, nvoxels


def _lattice_from_centers(centers):
    """Recognise a getCenters-style lattice in an explicit (V,3) centre list.

    Returns (bb_min float64 (3,), nvoxels int (3,), voxelsize) when ``centers`` equals, to ~1e-9 A,
    ``bb_min + index*voxelsize`` in x-slowest / z-fastest order; else None.  Asked on every call that brings its own
    centres (``usercenters``, and every call through ``install()``: the reference computes the centres and hands over a
    copy), so it is done in the library: ~10 us for a 24^3 grid.  (A memo of the last verdicts, keyed by an exact
    comparison with a kept copy, was tried first: the comparison alone costs as much as the native recognition.)"""
    c = np.asarray(centers, dtype=np.float64)
    if c.ndim != 2 or c.shape[1] != 3 or c.shape[0] < 2:
        return None
    return _recognise_lattice(c)


def _recognise_lattice(c):
    """The recognition itself, in the library (two passes over the array, ~10 us for a 24^3 grid; the numpy form below --
    kept as the specification the tests compare it with -- takes a dozen temporaries and ~120 us)."""
    from . import _lib
    c = np.ascontiguousarray(c, dtype=np.float64)
    bb_min = np.zeros(3, dtype=np.float64)
    nvox = np.zeros(3, dtype=np.int32)
    vs = ctypes.c_double(0.0)
    if not _lib.load().mkamd_lattice_from_centers(_lib._ptr(c), int(c.shape[0]), _lib._ptr(bb_min), _lib._ptr(nvox), ctypes.byref(vs)):
        return None
    return bb_min, nvox.astype(np.int64), float(vs.value)


def _recognise_lattice_numpy(c):
    V = c.shape[0]
    if c.ndim != 2 or c.shape[1] != 3 or V < 2:
        return None
    z = c[:, 2]
    breaks = np.nonzero(z[1:] <= z[:-1])[0]
    nz = int(breaks[0]) + 1 if breaks.size else V
    if V % nz:
        return None
    y = c[::nz, 1]
    breaks = np.nonzero(y[1:] <= y[:-1])[0]
    ny = int(breaks[0]) + 1 if breaks.size else y.shape[0]
    if (V // nz) % ny:
        return None
    nx = V // (nz * ny)
    steps = []
    if nz > 1:
        steps.append(c[1, 2] - c[0, 2])
    if ny > 1:
        steps.append(c[nz, 1] - c[0, 1])
    if nx > 1:
        steps.append(c[nz * ny, 0] - c[0, 0])
    if not steps or not all(s > 0 for s in steps):
        return None
    vs = float(steps[0])
    if any(abs(s - vs) > 1e-9 * max(1.0, abs(vs)) for s in steps):
        return None
    rebuilt = (_getGridCenters(nx, ny, nz, vs) + c[0]).reshape(V, 3)
    tol = 1e-9 * max(1.0, float(np.abs(c).max()))
    if np.abs(rebuilt - c).max() > tol:
        return None
    return c[0].copy(), np.array([nx, ny, nz]), vs


def _occupancyLatticeBegin(coords, channelsigmas, lattice):
    """First half of ``_getOccupancyC`` for a lattice this module built itself: inputs checked and shipped, kernels
    enqueued; the function returned waits and hands back the float64 [V, C] features."""
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    channelsigmas = np.ascontiguousarray(channelsigmas, dtype=np.float64)
    if coords.ndim != 2 or coords.shape[1] != 3:
        raise ValueError("coords and centers must be (n, 3) arrays")
    if channelsigmas.ndim != 2 or channelsigmas.shape[0] != coords.shape[0]:
        raise ValueError("channel sigmas must be (natoms, nchannels)")
    bb_min, nvoxels, voxelsize = lattice
    offs = np.array([0, coords.shape[0]], dtype=np.int64)
    end = _batch.voxelize_lattice_begin(coords, offs, channelsigmas, np.asarray(bb_min, np.float64)[None, :], nvoxels, voxelsize,
                                        dtype=np.float64)
    return lambda: end()[0]


def _getOccupancyCPU(coords, centers, channelsigmas):
    """``_getOccupancyC`` (voxeldescriptors.py:515-533) on the host: float64 [V, C], bit-identical to the reference."""
    from .occupancy_utils import calculate_occupancy_cpu
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    centers = np.ascontiguousarray(centers, dtype=np.float64)
    channelsigmas = np.ascontiguousarray(channelsigmas, dtype=np.float64)
    if coords.ndim != 2 or coords.shape[1] != 3 or centers.ndim != 2 or centers.shape[1] != 3:
        raise ValueError("coords and centers must be (n, 3) arrays")
    if channelsigmas.ndim != 2 or channelsigmas.shape[0] != coords.shape[0]:
        raise ValueError("channel sigmas must be (natoms, nchannels)")
    occupancies = np.zeros((centers.shape[0], channelsigmas.shape[1]), dtype=np.float64)
    calculate_occupancy_cpu(centers, coords, channelsigmas, occupancies)
    return occupancies


_LAST_LATTICE = {}          # number of centres -> (nvoxels, voxelsize) of the last lattice recognised in an array of that length
_GUESS_HOLD = {}       # length -> calls still to go without a guess (after mis-guesses)
_GUESS_MISSES = {}     # length -> mis-guesses in a row


def _getOccupancyC(coords, centers, channelsigmas, _lattice=None):
    """Occupancy of every (centre, channel): float64 [V, C] -- the job of the reference's
    ``_getOccupancyC`` (voxeldescriptors.py:515-533), executed by the HIP kernels.

    Lattice centres (known from getCenters, or recognised in the array) take the tiled lattice
    kernel; arbitrary centres take the explicit-centre kernel."""
    # (the reference copies with astype, :519-521; nothing here writes to the inputs, so arrays that already have the
    #  dtype and layout are used as they are -- the float64 centres of a 64^3 grid alone are 6 MB)
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    centers = np.asarray(centers)
    if _lattice is None:
        centers = np.ascontiguousarray(centers, dtype=np.float64)
    channelsigmas = np.ascontiguousarray(channelsigmas, dtype=np.float64)
    if coords.ndim != 2 or coords.shape[1] != 3 or centers.ndim != 2 or centers.shape[1] != 3:
        raise ValueError("coords and centers must be (n, 3) arrays")
    if channelsigmas.ndim != 2 or channelsigmas.shape[0] != coords.shape[0]:
        raise ValueError("channel sigmas must be (natoms, nchannels)")
    if _lattice is None and centers.shape[0] >= 2:
        # The caller brought the centres (`usercenters`; every call through install(): the reference computes them and
        # hands over a copy).  Whether they are a getCenters lattice is asked on every call (~10 us for a 24^3 grid) -- but
        # the answer is nearly always what it was for the last array of this length, shifted to this array's first centre.
        # So the kernels are started on that guess, the centres are checked WHILE the device computes, and only a wrong
        # guess (a different grid of the same size, centres that are no lattice) is thrown away and done again.
        nkey = centers.shape[0]
        guess = _LAST_LATTICE.get(nkey)
        finish = None
        with _CENTERS_LOCK:
            hold = _GUESS_HOLD.get(nkey, 0)
            if hold > 0:                                   # this length mis-guessed lately (grids of the same size that alternate,
                _GUESS_HOLD[nkey] = hold - 1               # steps that differ in the last bit): no guessing for a while
                guess = None
        if guess is not None:
            first = centers[0].copy()
            try:
                finish = _occupancyLatticeBegin(coords, channelsigmas, (first, guess[0], guess[1]))
            except Exception:                              # noqa: BLE001 -- the guess is only a guess: the plain way below
                finish = None
        if finish is not None:
            rec_err, features = None, None
            try:
                lattice = _recognise_lattice(centers)
            except Exception as e:                         # noqa: BLE001 -- raised below, once the speculative call has been collected
                lattice, rec_err = None, e
            try:
                features = finish()
            except Exception:                              # noqa: BLE001 -- a failed GUESS is not this call's failure: the plain way below
                features = None
            if rec_err is not None:
                raise rec_err
            if (features is not None and lattice is not None and np.array_equal(lattice[0], first)
                    and np.array_equal(lattice[1], guess[0]) and lattice[2] == guess[1]):
                with _CENTERS_LOCK:
                    _GUESS_MISSES.pop(nkey, None)
                return features
            del features                                   # a wrong guess: the call below does it with what was found; every
            with _CENTERS_LOCK:                            # further miss in a row doubles the calls this length goes unguessed
                _LAST_LATTICE.pop(nkey, None)
                m = _GUESS_MISSES[nkey] = min(_GUESS_MISSES.get(nkey, 0) + 1, 6)
                _GUESS_HOLD[nkey] = (1 << m) - 2           # 0, 2, 6, 14, 30, 62 calls
        else:
            lattice = _recognise_lattice(centers)
        if lattice is not None:
            with _CENTERS_LOCK:                            # (shared between the threads that use the drop-in API)
                _LAST_LATTICE[centers.shape[0]] = (np.asarray(lattice[1]).copy(), lattice[2])
                while len(_LAST_LATTICE) > 8:
                    _LAST_LATTICE.pop(next(iter(_LAST_LATTICE)), None)
    else:
        lattice = _lattice
    if lattice is not None:
        bb_min, nvoxels, voxelsize = lattice
        offs = np.array([0, coords.shape[0]], dtype=np.int64)
        V = int(np.prod(np.asarray(nvoxels, dtype=np.int64)))
        out = np.empty((1, V, channelsigmas.shape[1]), dtype=np.float64)     # the library widens while copying out
        return _batch.voxelize_lattice(coords, offs, channelsigmas, np.asarray(bb_min, np.float64)[None, :],
                                       nvoxels, voxelsize, out=out)[0]
    feats = _batch.occupancy_centers(centers, coords, channelsigmas)
    return feats.astype(np.float64)


def install(distances: bool = True):
    """Make an installed moleculekit use the MI355X kernels: swaps
    ``moleculekit.tools.voxeldescriptors._getOccupancyC`` (the sole caller of the Cython kernel,
    voxeldescriptors.py:356) for this module's and -- unless ``distances=False`` -- the functions of
    ``moleculekit.distance_utils`` for ``moleculekit_amd.distance_utils``' (what MetricDistance, ``calculate_contacts``,
    ``cdist`` / ``pdist`` and ``_detectCollisions`` import at call time).  Returns the original ``_getOccupancyC``;
    ``uninstall()`` puts everything back."""
    import moleculekit.tools.voxeldescriptors as ref

    if distances:
        try:
            import moleculekit.distance_utils  # noqa: F401
        except ImportError:                                             # (an installation without the compiled module)
            pass
        else:
            from . import distance_utils as _du
            _du.install()
    if getattr(ref, "_getOccupancyC_reference", None) is not None:      # already installed
        return ref._getOccupancyC_reference
    original = ref._getOccupancyC
    ref._getOccupancyC = lambda coords, centers, channelsigmas: _getOccupancyC(coords, centers, channelsigmas)
    ref._getOccupancyC_reference = original
    return original


def uninstall():
    """Undo ``install()``."""
    import sys
    import moleculekit.tools.voxeldescriptors as ref

    original = getattr(ref, "_getOccupancyC_reference", None)
    if original is not None:
        ref._getOccupancyC = original
        ref._getOccupancyC_reference = None
    if "moleculekit.distance_utils" in sys.modules:
        from . import distance_utils as _du
        _du.uninstall()
