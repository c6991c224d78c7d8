"""GPU drop-ins for the functions of ``moleculekit.distance_utils`` (Cython, distance_utils.pyx:59-434).

Same names, argument order, dtypes and in-place ``results`` semantics as the reference's typed-memoryview
functions; float32 results are BIT-EXACT with the reference (the kernels reproduce its float32 operation
order with un-contracted round-to-nearest ops, see csrc/dist_kernels.h).  The all-pairs distance work runs
in HIP kernels; the variable-length index lists of ``contacts_trajectory`` / ``get_collisions`` are thresholded
(``dist2 <= threshold^2`` in float32, as :82-90 and :111-120 do) and compacted on the GPU as well, in the
reference's loop order and in chunks of frames, so that no [frames x pairs] matrix ever exists on either side.
"""
from __future__ import annotations

import numpy as np

from . import _lib

FLOAT32, UINT32 = np.float32, np.uint32


def _req(name, a, dtype, ndim):
    if not isinstance(a, np.ndarray):
        raise TypeError(f"{name}: a numpy array is required")
    if a.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch for {name}: expected {np.dtype(dtype).name}, got {a.dtype.name}")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions for {name} (expected {ndim}, got {a.ndim})")
    return np.ascontiguousarray(a)


def _csr(groups):
    offs = np.zeros(len(groups) + 1, dtype=np.int64)
    if len(groups):
        offs[1:] = np.cumsum([len(g) for g in groups])
    atoms = np.fromiter((a for g in groups for a in g), dtype=np.int32, count=int(offs[-1]))
    return atoms, offs


def _store(results, tmp):
    if results.shape != tmp.shape:
        raise ValueError(f"results must have shape {tmp.shape}, got {results.shape}")
    if tmp is not results:
        results[...] = tmp


def _target(results, shape):
    """Where the library writes: the caller's array itself when it is a C-contiguous float32 array of the right shape
    (no second pass over hundreds of megabytes), else a temporary that _store copies from."""
    if results.shape != shape:
        raise ValueError(f"results must have shape {shape}, got {results.shape}")
    if results.dtype == np.float32 and results.flags["C_CONTIGUOUS"] and results.flags["WRITEABLE"]:
        return results
    return np.empty(shape, dtype=np.float32)


def _frame_inputs(coords, box, chains, pbc, used_atoms):
    """What the library reads wholesale -- 3 x F box floats and one chain id per atom -- from what the reference reads
    only where it needs it: the box only when ``pbc`` (distance_utils.pyx:49-52), a chain id only at the atoms the call
    touches.  A box of another shape is fine without pbc (zeros stand in), an error with it; a short chain array is
    padded once every touched atom has an entry."""
    N, F = coords.shape[0], coords.shape[2]
    if coords.shape[1] != 3:
        raise ValueError(f"coords must be (natoms, 3, nframes), got {coords.shape}")
    if box.shape != (3, F):
        if pbc:
            raise ValueError(f"box must have shape (3, {F}) for periodic distances, got {box.shape}")
        box = np.zeros((3, F), dtype=np.float32)
    if chains.shape[0] < N:
        top = int(max((int(np.max(u)) for u in used_atoms if len(u)), default=-1))
        if top >= chains.shape[0]:
            raise ValueError(f"digitized_chains has {chains.shape[0]} entries but atom {top} is used")
        chains = np.concatenate([chains, np.zeros(N - chains.shape[0], dtype=np.uint32)])
    return box, chains


def dist_trajectory(coords, box, sel1, sel2, digitized_chains, selfdist, pbc, results, ctx=None):
    """distance_utils.pyx:126-155: ``results[f, idx] = |coords[sel1[i],:,f] - coords[sel2[j],:,f]|`` with the
    orthorhombic minimum image across different chains when ``pbc``."""
    coords = _req("coords", coords, np.float32, 3)
    box = _req("box", box, np.float32, 2)
    sel1 = _req("sel1", sel1, np.uint32, 1); sel2 = _req("sel2", sel2, np.uint32, 1)
    chains = _req("digitized_chains", digitized_chains, np.uint32, 1)
    _req("results", results, np.float32, 2)
    box, chains = _frame_inputs(coords, box, chains, pbc, (sel1, sel2))
    ctx = ctx or _lib.default_context()
    F = coords.shape[2]
    npairs = int(_lib.load().mkamd_dist_count_pairs(len(sel1), len(sel2), int(bool(selfdist))))
    tmp = _target(results, (F, npairs))
    ctx.dist_trajectory_host(coords, box, sel1, sel2, chains, bool(selfdist), bool(pbc), False, tmp)
    _store(results, tmp)


def contacts_trajectory(coords, box, sel1, sel2, digitized_chains, selfdist, pbc, dist_threshold=5, ctx=None):
    """distance_utils.pyx:59-93: per frame the flat list ``[a0, b0, a1, b1, ...]`` of atom pairs with
    ``dist2 <= threshold^2`` (float32), in the reference's (i, j) loop order."""
    coords = _req("coords", coords, np.float32, 3)
    box = _req("box", box, np.float32, 2)
    sel1 = _req("sel1", sel1, np.uint32, 1); sel2 = _req("sel2", sel2, np.uint32, 1)
    chains = _req("digitized_chains", digitized_chains, np.uint32, 1)
    box, chains = _frame_inputs(coords, box, chains, pbc, (sel1, sel2))
    ctx = ctx or _lib.default_context()
    offs, pairs = ctx.contacts_trajectory_host(coords, box, sel1, sel2, chains, bool(selfdist), bool(pbc), dist_threshold)
    flat = pairs.astype(np.int64).ravel()
    return [flat[2 * offs[f]:2 * offs[f + 1]].tolist() for f in range(coords.shape[2])]


def get_collisions(coords1, coords2, dist_threshold, ctx=None):
    """distance_utils.pyx:98-121: flat ``[i0, j0, i1, j1, ...]`` (row indices) with dist2 <= threshold^2."""
    c1 = _req("coords1", coords1, np.float32, 2); c2 = _req("coords2", coords2, np.float32, 2)
    # the one-frame, non-periodic case of the contact kernels on the concatenation of the two sets
    n1, n2 = c1.shape[0], c2.shape[0]
    both = np.ascontiguousarray(np.concatenate([c1[:, :3], c2[:, :3]])[:, :, None])
    sel1 = np.arange(n1, dtype=np.uint32); sel2 = np.arange(n1, n1 + n2, dtype=np.uint32)
    _, pairs = (ctx or _lib.default_context()).contacts_trajectory_host(
        both, np.zeros((3, 1), np.float32), sel1, sel2, np.zeros(n1 + n2, np.uint32), False, False, dist_threshold)
    flat = pairs.astype(np.int64)
    flat[:, 1] -= n1
    return flat.ravel().tolist()


def _reduction(coords, box, groups1, groups2, ch1, ch2, selfdist, pairs, pbc, masses, r1, r2, results, ctx):
    coords = _req("coords", coords, np.float32, 3)
    box = _req("box", box, np.float32, 2)
    ch1 = _req("digitized_chains1", ch1, np.uint32, 1); ch2 = _req("digitized_chains2", ch2, np.uint32, 1)
    masses = _req("masses", masses, np.float32, 1)
    _req("results", results, np.float32, 2)
    a1, o1 = _csr(groups1); a2, o2 = _csr(groups2)
    box, _ = _frame_inputs(coords, box, np.zeros(coords.shape[0], np.uint32), pbc, ())
    # (chain ids are per GROUP here, distance_utils.pyx:225-229: the library reads one per group)
    if ch1.shape[0] < len(groups1) or ch2.shape[0] < len(groups2):
        raise ValueError(f"digitized_chains1/2 have {ch1.shape[0]}/{ch2.shape[0]} entries for {len(groups1)}/{len(groups2)} groups")
    if a1.size and (a1.min() < 0 or a1.max() >= coords.shape[0]) or a2.size and (a2.min() < 0 or a2.max() >= coords.shape[0]):
        raise ValueError("group atom index out of range")
    if masses.shape[0] < coords.shape[0]:            # read only by the centre-of-mass reduction (distance_utils.pyx:158-185)
        if int(r1) == 1 or int(r2) == 1:
            raise ValueError(f"masses has {masses.shape[0]} entries for {coords.shape[0]} atoms")
        masses = np.concatenate([masses, np.zeros(coords.shape[0] - masses.shape[0], dtype=np.float32)])
    ctx = ctx or _lib.default_context()
    F = coords.shape[2]
    nout = len(groups1) if pairs else int(_lib.load().mkamd_dist_count_pairs(len(groups1), len(groups2), int(bool(selfdist))))
    tmp = _target(results, (F, nout))
    ctx.dist_reduction_host(coords, box, a1, o1, a2, o2, ch1, ch2, bool(selfdist), bool(pairs), bool(pbc), masses,
                            int(r1), int(r2), tmp)
    _store(results, tmp)
    return results


def dist_trajectory_reduction(coords, box, groups1, groups2, digitized_chains1, digitized_chains2, selfdist, pbc,
                              masses, reduction1, reduction2, results, ctx=None):
    """distance_utils.pyx:211-281: group-vs-group minimum ("closest", 0) or centre-of-mass (1) distances."""
    return _reduction(coords, box, groups1, groups2, digitized_chains1, digitized_chains2, selfdist, False, pbc,
                      masses, reduction1, reduction2, results, ctx)


def dist_trajectory_reduction_pairs(coords, box, groups1, groups2, digitized_chains1, digitized_chains2, pbc, masses,
                                    reduction1, reduction2, results, ctx=None):
    """distance_utils.pyx:286-350: as above for the pairs (groups1[g], groups2[g])."""
    return _reduction(coords, box, groups1, groups2, digitized_chains1, digitized_chains2, False, True, pbc, masses,
                      reduction1, reduction2, results, ctx)


def cdist(coords1, coords2, results, ctx=None):
    """distance_utils.pyx:355-383."""
    c1 = _req("coords1", coords1, np.float32, 2); c2 = _req("coords2", coords2, np.float32, 2)
    _req("results", results, np.float32, 2)
    if c1.shape[1] != c2.shape[1]:
        raise ValueError("Second dimension of input arguments must match")
    tmp = _target(results, (c1.shape[0], c2.shape[0]))
    (ctx or _lib.default_context()).cdist_host(c1, c2, tmp)
    _store(results, tmp)


def pdist(coords, results, ctx=None):
    """distance_utils.pyx:388-416."""
    c = _req("coords", coords, np.float32, 2)
    _req("results", results, np.float32, 1)
    n = c.shape[0]
    tmp = _target(results, (n * (n - 1) // 2,))
    (ctx or _lib.default_context()).pdist_host(c, tmp)
    _store(results, tmp)


def squareform(distances):
    """distance_utils.pyx:421-434: condensed vector -> symmetric matrix (pure index work)."""
    d = np.ascontiguousarray(distances, dtype=np.float32)
    n = d.shape[0]
    newdim = int((np.sqrt(8 * n + 1) + 1) / 2)
    out = np.zeros((newdim, newdim), dtype=np.float32)
    iu = np.triu_indices(newdim, k=1)
    out[iu] = d[: len(iu[0])]
    out[(iu[1], iu[0])] = d[: len(iu[0])]
    return out


# ------------------------------------------------------------------------------------------------
# The drop-in hook for this row.  The reference reaches its compiled ``moleculekit.distance_utils`` only through
# function-local imports -- projections/util.py:22 (pp_calcDistances) and :100 (get_reduced_distances) behind MetricDistance,
# distance.py:242 / :278 / :308 / :360 (cdist, pdist, squareform, calculate_contacts), molecule.py:3731 (_detectCollisions) --
# so swapping the attributes of that module makes every one of those callers run on the GPU unchanged.
# ------------------------------------------------------------------------------------------------
HOOKED = ("dist_trajectory", "contacts_trajectory", "get_collisions", "dist_trajectory_reduction",
          "dist_trajectory_reduction_pairs", "cdist", "pdist", "squareform")


def install():
    """Swap the eight functions of an installed ``moleculekit.distance_utils`` for this module's (same positional
    signatures, in-place ``results``).  Returns ``{name: original}``; idempotent; ``uninstall()`` puts them back."""
    import moleculekit.distance_utils as ref

    saved = getattr(ref, "_mkamd_reference", None)
    if saved is not None:
        return saved
    mine = globals()
    saved = {name: getattr(ref, name) for name in HOOKED if hasattr(ref, name)}
    for name in saved:
        setattr(ref, name, mine[name])
    ref._mkamd_reference = saved
    return saved


def uninstall():
    """Undo ``install()``."""
    import moleculekit.distance_utils as ref

    saved = getattr(ref, "_mkamd_reference", None)
    if saved is not None:
        for name, fn in saved.items():
            setattr(ref, name, fn)
        ref._mkamd_reference = None
