"""Small host-side helpers beside the voxel-descriptor path.

``boundingBox`` mirrors moleculekit/util.py:354-379 and ``rotationMatrix`` moleculekit/util.py:70-118
(the pieces ``getCenters`` / ``rotateCoordinates`` need).  Molecules are duck-typed: anything with
``get("coords")`` (current frame, ``(natoms, 3)``) works; a plain ``(natoms, 3)`` array is accepted too.
"""
from __future__ import annotations

import math

import numpy as np


def _frame_coords(mol, sel="all"):
    """Current-frame coordinates of a duck-typed molecule (molecule.py:679-683 semantics)."""
    if isinstance(mol, np.ndarray):
        cc = mol
    elif hasattr(mol, "get"):
        try:
            cc = mol.get("coords", sel=sel) if sel not in (None, "all") else mol.get("coords")
        except TypeError:
            cc = mol.get("coords")
    elif hasattr(mol, "coords"):
        cc = np.asarray(mol.coords)
        if cc.ndim == 3:
            cc = cc[:, :, getattr(mol, "frame", 0)]
    else:
        raise TypeError("mol must provide get('coords') or .coords")
    cc = np.asarray(cc)
    if cc.ndim == 3:
        cc = np.squeeze(cc, axis=2) if cc.shape[2] == 1 else cc[:, :, 0]
    if cc.ndim == 1:
        cc = cc[np.newaxis, :]
    return cc


def boundingBox(mol, sel="all") -> np.ndarray:
    """(2, 3) array ``[min, max]`` over the selected atoms' coordinates, in the coordinates' own
    dtype (float32 for a Molecule) -- util.py:376-379."""
    coords = _frame_coords(mol, sel)
    return np.vstack((np.squeeze(np.min(coords, axis=0)), np.squeeze(np.max(coords, axis=0))))


def rotationMatrix(axis, theta) -> np.ndarray:
    """Counter-clockwise rotation by ``theta`` radians about ``axis`` (Euler-Rodrigues form,
    util.py:70-118)."""
    axis = np.asarray(axis, dtype=float)
    theta = np.asarray(theta)
    axis = axis / math.sqrt(np.dot(axis, axis))
    a = math.cos(theta / 2)
    b, c, d = -axis * math.sin(theta / 2)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([
        [aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)],
        [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)],
        [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc],
    ])


# ------------------------------------------------------------------------------------------------
# .cube export / import of voxel grids (moleculekit/util.py:415-501)
# ------------------------------------------------------------------------------------------------
_BOHR = 0.52917725          # the constant the reference converts with (util.py:432)


def writeCube(arr, filename, vecMin, vecRes):
    """Write a 3-D array as a Gaussian .cube file -- byte-identical to the reference's ``writeCube``
    (util.py:415-458): header in Bohr with the origin at the first voxel's centre (``vecMin + vecRes/2``), then
    the values x-outer / z-inner, ``%13.5g`` each, six per line.  (Vectorised: the reference formats value by
    value in a triple Python loop.)"""
    arr = np.asarray(arr)
    if arr.ndim != 3:
        raise ValueError("writeCube needs a 3-D array")
    vecRes = np.array(vecRes, dtype=np.float64)
    vecMin = np.array(vecMin, dtype=np.float64)
    L = 1 / _BOHR
    gauss_bin = vecRes * L
    minCorner = L * (vecMin + 0.5 * vecRes)
    ngrid = arr.shape
    vals = ["%13.5g" % v for v in arr.ravel().tolist()]
    lines = ["".join(vals[i:i + 6]) for i in range(0, len(vals), 6)]
    body = "\n".join(lines)
    if len(vals) % 6 == 0 and vals:
        body += "\n"                                 # the reference ends a full last line with a newline, a partial one not
    with open(filename, "w") as f:
        f.write("CUBE FILE\n")
        f.write("OUTER LOOP: X, MIDDLE LOOP: Y, INNER LOOP: Z\n")
        f.write("%5d %12.6f %12.6f %12.6f\n" % (1, minCorner[0], minCorner[1], minCorner[2]))
        f.write("%5d %12.6f %12.6f %12.6f\n" % (ngrid[0], gauss_bin[0], 0, 0))
        f.write("%5d %12.6f %12.6f %12.6f\n" % (ngrid[1], 0, gauss_bin[1], 0))
        f.write("%5d %12.6f %12.6f %12.6f\n" % (ngrid[2], 0, 0, gauss_bin[2]))
        f.write("%5d %12.6f %12.6f %12.6f %12.6f\n" % (1, 0, minCorner[0], minCorner[1], minCorner[2]))
        f.write(body)


def readCube(fname):
    """Read a .cube file back: ``(data float64 [nx,ny,nz], meta)`` like util.py:461-501."""
    meta = {}
    with open(fname) as f:
        f.readline(); f.readline()

        def get():
            ll = f.readline().split()
            return int(ll[0]), [float(x) for x in ll[1:]]
        natm, meta["org"] = get()
        nx, meta["xvec"] = get()
        ny, meta["yvec"] = get()
        nz, meta["zvec"] = get()
        meta["atoms"] = [get() for _ in range(natm)]
        data = np.array(f.read().split(), dtype=np.float64)
    return data.reshape(nx, ny, nz), meta


def writeVoxelFeatures(features, centers, nvoxels, prefix, voxelsize=None, featurenames=None):
    """One .cube file per channel of a ``getVoxelDescriptors`` result (``features [V, C]``, ``centers [V, 3]``,
    ``nvoxels``): the volumes the reference's ``viewVoxelFeatures`` (tools/voxeldescriptors.py:30-75) hands to VMD,
    with its conventions (voxel size inferred from the first two centres, lower edge = min centre - voxelsize/2).
    Returns the file names ``<prefix>_<name>.cube``."""
    from .channels import CHANNEL_ORDER
    features, centers = np.asarray(features), np.asarray(centers)
    nv = [int(v) for v in nvoxels]
    names = list(featurenames) if featurenames is not None else list(CHANNEL_ORDER)[:features.shape[1]]
    if voxelsize is None:
        voxelsize = np.repeat(abs(centers[0, 2] - centers[1, 2]), 3)
    voxelsize = np.array(voxelsize, dtype=np.float64)
    vol = features.reshape(nv + [features.shape[1]])
    loweredge = np.min(centers.reshape(nv + [3]), axis=(0, 1, 2)) - voxelsize / 2
    out = []
    for i, name in enumerate(names):
        fn = f"{prefix}_{name}.cube"
        writeCube(vol[..., i], fn, loweredge, voxelsize)
        out.append(fn)
    return out
