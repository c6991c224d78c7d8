"""Host wrappers of moleculekit/distance.py:221-412 on top of the GPU ``distance_utils`` (same signatures):
``cdist``, ``pdist``, ``squareform``, ``calculate_contacts``; plus the drivers of
moleculekit/projections/util.py:12-223 (``pp_calcDistances``, ``get_reduced_distances``) that MetricDistance uses.
Molecules are duck-typed (``coords`` [N,3,F] float32, ``box`` [3,F], ``chain``, ``element``, ``numAtoms``, ``numFrames``)."""
from __future__ import annotations

import numpy as np

from . import distance_utils as _du


def cdist(coords1, coords2):
    """(N, D) x (M, D) -> float32 (N, M) Euclidean distances (distance.py:221-252)."""
    coords1, coords2 = np.asarray(coords1), np.asarray(coords2)
    assert coords1.ndim == 2, "cdist only supports 2D arrays"
    assert coords2.ndim == 2, "cdist only supports 2D arrays"
    assert coords1.shape[1] == coords2.shape[1], "Second dimension of input arguments must match"
    results = np.zeros((coords1.shape[0], coords2.shape[0]), dtype=np.float32)
    _du.cdist(coords1.astype(np.float32), coords2.astype(np.float32), results)
    return results


def pdist(coords):
    """(N, D) -> condensed float32 upper-triangular distances (distance.py:255-282)."""
    coords = np.asarray(coords)
    assert coords.ndim == 2, "pdist only supports 2D arrays"
    n = coords.shape[0]
    results = np.zeros(int(n * (n - 1) / 2), dtype=np.float32)
    _du.pdist(coords.astype(np.float32), results)
    return results


def squareform(distances):
    """Condensed vector -> (N, N) symmetric matrix (distance.py:285-305)."""
    return np.array(_du.squareform(np.asarray(distances).astype(np.float32)))


def _periodic_setup(mol, sel2_atoms, periodic):
    """box + digitized chains exactly as projections/util.py:24-53 / distance.py:362-392 build them."""
    coords, box = mol.coords, mol.box
    if periodic is not None:
        if box is None or np.sum(box) == 0:
            raise RuntimeError(
                "No periodic box dimensions given in the molecule/trajectory. "
                "If you want to calculate distance without wrapping, set the periodic option to None")
    else:
        box = np.zeros((3, coords.shape[2]), dtype=np.float32)
    if box.shape[1] != coords.shape[2]:
        raise RuntimeError("Different number of frames in mol.coords and mol.box. "
                           "Please ensure they both have the same number of frames")
    natoms = coords.shape[0]
    if periodic is None:
        chains = np.zeros(natoms, dtype=np.uint32)
    elif periodic == "chains":
        chains = np.unique(mol.chain, return_inverse=True)[1].astype(np.uint32)
    elif periodic == "selections":
        chains = np.ones(natoms, dtype=np.uint32)
        chains[sel2_atoms] = 2
    else:
        raise RuntimeError(f"Invalid periodic option {periodic}")
    return np.ascontiguousarray(coords, np.float32), np.ascontiguousarray(box, np.float32), chains


def calculate_contacts(mol, sel1, sel2, periodic, threshold=4):
    """Per frame the (n, 2) uint32 atom-index pairs within ``threshold`` (distance.py:308-412)."""
    assert isinstance(sel1, np.ndarray) and sel1.dtype == bool
    assert isinstance(sel2, np.ndarray) and sel2.dtype == bool
    selfdist = np.array_equal(sel1, sel2)
    sel1 = np.where(sel1)[0].astype(np.uint32)
    sel2 = np.where(sel2)[0].astype(np.uint32)
    coords, box, chains = _periodic_setup(mol, sel2, periodic)
    res = _du.contacts_trajectory(coords, box, sel1, sel2, chains, selfdist, periodic is not None, threshold)
    return [np.array(r, dtype=np.uint32).reshape(-1, 2) for r in res]


def pp_calcDistances(mol, sel1, sel2, periodic, metric="distances", threshold=8, gap=1, truncate=None):
    """Atom-vs-atom distances of every frame: float32 (numFrames, npairs) (projections/util.py:12-85)."""
    selfdist = np.array_equal(sel1, sel2)
    sel1 = np.where(sel1)[0].astype(np.uint32)
    sel2 = np.where(sel2)[0].astype(np.uint32)
    coords, box, chains = _periodic_setup(mol, sel2, periodic)
    F = coords.shape[2]
    shape = (F, len(sel1) * len(sel2))
    if selfdist:
        shape = (F, int((len(sel1) * (len(sel2) - 1)) / 2))
    results = np.zeros(shape, dtype=np.float32)
    _du.dist_trajectory(coords, box, sel1, sel2, chains, selfdist, periodic is not None, results)
    if truncate is not None:
        results[results > truncate] = truncate
    if metric == "contacts":
        results = results <= threshold
    elif metric != "distances":
        raise RuntimeError("The metric you asked for is not supported. Check spelling and documentation")
    return results


def get_reduced_distances(mol, sel1, sel2, periodic, metric="distances", threshold=8, truncate=None,
                          reduction1="closest", reduction2="closest", pairs=False, masses=None):
    """Group-vs-group distances (closest atom pair or centres of mass) of every frame
    (projections/util.py:88-223). ``masses`` defaults to the element masses of ``mol.element``."""
    sel1, sel2 = np.asarray(sel1), np.asarray(sel2)
    if np.ndim(sel1) != 2:
        idx = np.where(sel1)[0]
        g = np.zeros((len(idx), len(sel1)), dtype=bool); g[np.arange(len(idx)), idx] = True; sel1 = g
    if np.ndim(sel2) != 2:
        idx = np.where(sel2)[0]
        g = np.zeros((len(idx), len(sel2)), dtype=bool); g[np.arange(len(idx)), idx] = True; sel2 = g
    selfdist = np.array_equal(sel1, sel2)
    sel2_atoms = np.where(sel2.any(axis=0))[0]
    coords, box, chains = _periodic_setup(mol, sel2_atoms, periodic)
    groups1 = [np.where(sel1[i, :])[0].tolist() for i in range(sel1.shape[0])]
    groups2 = [np.where(sel2[i, :])[0].tolist() for i in range(sel2.shape[0])]
    if pairs and len(groups1) != len(groups2):
        raise RuntimeError("If `pairs=True` mode is used, the number of groups in sel1 should match the number of groups in sel2.")
    F = coords.shape[2]
    if selfdist:
        mindist = np.zeros((F, int((len(groups1) * (len(groups2) - 1)) / 2)), dtype=np.float32)
    elif not pairs:
        mindist = np.zeros((F, len(groups1) * len(groups2)), dtype=np.float32)
    else:
        mindist = np.zeros((F, len(groups1)), dtype=np.float32)
    rmap = {"closest": 0, "com": 1}
    ch1 = np.array([chains[g[0]] for g in groups1], dtype=np.uint32)
    ch2 = np.array([chains[g[0]] for g in groups2], dtype=np.uint32)
    if masses is None:
        from ._element_masses import ELEMENT_MASS
        masses = np.array([ELEMENT_MASS[el] for el in mol.element], dtype=np.float32)
    masses = np.ascontiguousarray(masses, dtype=np.float32)
    if not pairs:
        _du.dist_trajectory_reduction(coords, box, groups1, groups2, ch1, ch2, selfdist, periodic is not None, masses,
                                      rmap[reduction1.lower()], rmap[reduction2.lower()], mindist)
    else:
        _du.dist_trajectory_reduction_pairs(coords, box, groups1, groups2, ch1, ch2, periodic is not None, masses,
                                            rmap[reduction1.lower()], rmap[reduction2.lower()], mindist)
    if truncate is not None:
        mindist[mindist > truncate] = truncate
    if metric == "contacts":
        mindist = mindist <= threshold
    elif metric != "distances":
        raise RuntimeError("The metric you asked for is not supported. Check spelling and documentation")
    return mindist
