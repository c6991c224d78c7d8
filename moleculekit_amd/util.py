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
