"""``calculate_occupancy`` with the exact call contract of the reference's Cython kernel
(moleculekit/occupancy_utils/occupancy_utils.pyx:34-61), executed on the MI355X; ``calculate_occupancy_cpu``: the same
contract on the host (the library's own double-precision implementation, explicit only).

    calculate_occupancy(centers f64 [V,3], coords f32 [N,3], sigmas f64 [N,C], results f64 [V,C])

max-accumulates IN PLACE into ``results`` and returns ``None``.  Like the typed-memoryview
signature of the reference it raises ``ValueError`` on a dtype / ndim mismatch instead of converting.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _require(name, a, dtype, ndim=2):
    if not isinstance(a, np.ndarray):
        raise TypeError(f"{name}: a numpy array is required")
    if a.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch for {name}: expected {np.dtype(dtype).name}, got {a.dtype.name}")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions for {name} (expected {ndim}, got {a.ndim})")


def _checked(centers, coords, sigmas, results):
    _require("centers", centers, np.float64)
    _require("coords", coords, np.float32)
    _require("sigmas", sigmas, np.float64)
    _require("results", results, np.float64)
    V, N, C = centers.shape[0], coords.shape[0], sigmas.shape[1]
    if centers.shape[1] != 3 or coords.shape[1] != 3 or sigmas.shape[0] != N or results.shape != (V, C):
        raise ValueError("shape mismatch: centers [V,3], coords [N,3], sigmas [N,C], results [V,C]")
    return np.ascontiguousarray(centers), np.ascontiguousarray(coords), np.ascontiguousarray(sigmas)


def calculate_occupancy(centers, coords, sigmas, results, ctx=None):
    cen, xyz, sig = _checked(centers, coords, sigmas, results)
    ctx = ctx or _lib.default_context()
    if results.flags["C_CONTIGUOUS"]:
        ctx.calculate_occupancy(cen, xyz, sig, results)
    else:  # strided memoryviews are legal in the reference
        tmp = np.ascontiguousarray(results)
        ctx.calculate_occupancy(cen, xyz, sig, tmp)
        results[...] = tmp
    return None


def calculate_occupancy_cpu(centers, coords, sigmas, results, n_threads=0):
    """The same contract on the HOST (include/mkamd_voxel.h, mkamd_calculate_occupancy_cpu): the library's own
    double-precision implementation -- a cell list over the atoms, the reference's arithmetic per pair, bit-identical
    results -- for hosts without a GPU.  Explicit only: ``calculate_occupancy`` never falls back to it."""
    cen, xyz, sig = _checked(centers, coords, sigmas, results)
    V, N, C = cen.shape[0], xyz.shape[0], sig.shape[1]
    L = _lib.load_host()                     # libmkamd_host.so: plain C++, no ROCm needed (the copy inside libmkamd.so is for C callers)
    tmp = results if results.flags["C_CONTIGUOUS"] else np.ascontiguousarray(results)
    st = L.mkamd_calculate_occupancy_cpu_threads(_lib._ptr(cen), V, _lib._ptr(xyz), N, _lib._ptr(sig), C, _lib._ptr(tmp), int(n_threads))
    if st:
        msg = L.mkamd_host_last_error().decode(errors="replace")
        raise (ValueError if st == 1 else MemoryError if st == 6 else RuntimeError)(f"libmkamd_host: {msg}")
    if tmp is not results:
        results[...] = tmp
    return None
