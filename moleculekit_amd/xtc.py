"""XTC trajectory reading on top of libmkamd.so's host-side decoder.

Mirrors the reference's low-level module ``moleculekit.xtc`` (fileformats/xtc/xtc.pyx: ``get_xtc_natoms``,
``get_xtc_nframes``, ``read_xtc`` :34-53, ``read_xtc_frames`` :57-81 -- same arrays, same float32 bits, units as in the
file: nm / ps) and ``readers.XTCread`` (readers.py:1830-1866: the conversion to Angstrom / fs and from box vectors to
lengths and angles, unitcell.py:128-221).  Frames are decoded in parallel on host threads.  ``XTCread`` returns a
plain ``Trajectory`` record (coords float32 [N,3,F] in A, box [3,F], boxangles, step, time) -- the arrays
``batch.iterVoxelizeTrajectory`` / ``voxelizeTrajectory`` take; building a ``Molecule`` around it is outside this package.
"""
from __future__ import annotations

import ctypes
from collections import namedtuple

import numpy as np

from . import _lib

Trajectory = namedtuple("Trajectory", ["coords", "box", "boxangles", "step", "time"])


def _path(filename) -> bytes:
    return filename if isinstance(filename, bytes) else str(filename).encode("UTF-8")


def _info(filename):
    na, nf = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib._check(_lib.load().mkamd_xtc_info(_path(filename), ctypes.byref(na), ctypes.byref(nf)))
    return na.value, nf.value


def get_xtc_natoms(filename) -> int:
    return _info(filename)[0]


def get_xtc_nframes(filename) -> int:
    return _info(filename)[1]


def _read(filename, frames, nthreads):
    natoms, nframes = _info(filename)
    if frames is None:
        sel, n = None, nframes
    else:
        sel = np.ascontiguousarray(frames, dtype=np.int64).reshape(-1)
        n = len(sel)
    coords = np.zeros((natoms, 3, n), dtype=np.float32)
    box = np.zeros((3, 3, n), dtype=np.float32)
    time = np.zeros(n, dtype=np.float32)
    step = np.zeros(n, dtype=np.int32)
    _lib._check(_lib.load().mkamd_xtc_read(_path(filename), _lib._ptr(sel), n, natoms, _lib._ptr(coords), _lib._ptr(box),
                                           _lib._ptr(time), _lib._ptr(step), int(nthreads)))
    return coords, box, time, step


def read_xtc(filename, nthreads: int = 0):
    """All frames: ``(coords f32 [N,3,F] nm, boxvectors f32 [3,3,F] nm, time f32 [F] ps, step i32 [F])``."""
    return _read(filename, None, nthreads)


def read_xtc_frames(filename, frames, nthreads: int = 0):
    """The listed frames (any order), same return layout as ``read_xtc``."""
    return _read(filename, frames, nthreads)


def box_vectors_to_lengths_and_angles(a, b, c):
    """Box lengths and angles (degrees) from box vectors, per frame (unitcell.py:128-221); all-zero vectors give
    zero lengths and 90 degree angles."""
    a, b, c = (np.asarray(v) for v in (a, b, c))
    if np.all(a == 0) or np.all(b == 0) or np.all(c == 0):
        nf = a.shape[:-1]
        return np.zeros(nf), np.zeros(nf), np.zeros(nf), np.full(nf, 90.0), np.full(nf, 90.0), np.full(nf, 90.0)
    la, lb, lc = (np.sqrt(np.sum(v * v, axis=-1)) for v in (a, b, c))
    ang = lambda u, v, lu, lv: np.arccos(np.einsum("...i, ...i", u, v) / (lu * lv)) * 180.0 / np.pi
    return la, lb, lc, ang(b, c, lb, lc), ang(c, a, lc, la), ang(a, b, la, lb)


def XTCread(filename, frame=None, nthreads: int = 0) -> Trajectory:
    """``readers.XTCread`` (readers.py:1830-1866): coordinates and box in Angstrom (x10), time in fs (x1e3); step
    and time replaced by 0..F-1 / zeros when the file holds none (all-zero sums), like the reference."""
    if frame is None:
        coords, boxvectors, time, step = read_xtc(filename, nthreads)
    else:
        coords, boxvectors, time, step = read_xtc_frames(filename, np.atleast_1d(frame), nthreads)
    if coords.shape[2] == 0:
        raise RuntimeError(f"Malformed XTC file. No frames read from: {filename}")
    if coords.shape[0] == 0:
        raise RuntimeError(f"Malformed XTC file. No atoms read from: {filename}")
    time = time.astype(np.float64)
    coords *= 10.0
    boxvectors *= 10.0
    time *= 1e3
    nframes = coords.shape[2]
    if np.sum(step) == 0:
        step = np.arange(nframes, dtype=np.uint64)
    if np.sum(time) == 0:
        time = np.zeros(nframes, dtype=np.float64)
    bx, by, bz, alpha, beta, gamma = box_vectors_to_lengths_and_angles(boxvectors[0].T, boxvectors[1].T, boxvectors[2].T)
    return Trajectory(coords=coords, box=np.stack([bx, by, bz], axis=0), boxangles=np.stack([alpha, beta, gamma], axis=0),
                      step=step, time=time)
