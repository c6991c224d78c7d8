#!/usr/bin/env python3
"""Golden fixture for the distance_utils row: the KNOWN ANSWERS of the reference's own MetricDistance tests (SURVEY.md section 8f-1:
"pinned by tests/test_metricdistance.py:99-493 semantics").

The reference's tests/test_metricdistance.py builds tiny molecules with analytic answers (:99-181 distances, wrapping in a 2 A box,
the minimum over a group, the order of the pairs; :355-399 centre-of-mass / closest-atom reductions over residues), quotes four
numbers for 3PTB's protein against its benzamidine (:401-464: 8.978174, 3.8286476, 2.8153415 and the distance of the two centres of
mass), pairs mode (:467-493) and the three `periodic=` modes on its trajectory (:329-352).  This script drives the REAL reference
(built in a scratch directory, see make_golden.py) through those scenarios and records every call the projections make into
`moleculekit.distance_utils`: the arguments as the reference's drivers built them (coordinates and box included for the small
molecules; the trajectory's come from tests/golden/xtc/metricdistance_traj.xtc), the compiled reference's result, and the known
answer the reference's test asserts with its tolerance.  Run in the build container only; nothing of the reference travels:

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_metricdistance_known.py

Stores DATA only: tests/golden/metricdistance_known.npz.
"""
import os
import sys

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
REF_TESTS = os.environ.get("MOLECULEKIT_REF_TESTS", "/root/reference/tests")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)
import moleculekit.distance_utils as du  # noqa: E402
from moleculekit.molecule import Molecule  # noqa: E402
from moleculekit.projections.metricdistance import MetricDistance  # noqa: E402

CALLS = []


def _spy(name):
    real = getattr(du, name)

    def wrapper(*args):
        out = real(*args)
        CALLS.append((name, args[:-1] + (np.array(args[-1], copy=True),)))      # `results` as the compiled function left it (the drivers clip in place)
        return out

    setattr(du, name, wrapper)        # the drivers import the function at call time (projections/util.py:22,100)


def _csr(groups):
    offs = np.zeros(len(groups) + 1, np.int64)
    offs[1:] = np.cumsum([len(g) for g in groups])
    return np.concatenate([np.asarray(g, np.int32) for g in groups]), offs


OUTD = {}
NAMES = []


def record(key, metr, mol, known=None, tol=None, how="allclose", coords_tag=None):
    """Project, keep the ONE distance_utils call it made and the reference test's assertion about the projection."""
    CALLS.clear()
    data = np.asarray(metr.project(mol), np.float32)
    assert len(CALLS) == 1, (key, [c[0] for c in CALLS])
    name, a = CALLS[0]
    o = OUTD
    o[f"{key}_fn"] = np.array(name)
    if name == "dist_trajectory":
        coords, box, sel1, sel2, chains, selfdist, pbc, results = a
        o[f"{key}_sel1"], o[f"{key}_sel2"], o[f"{key}_chains"] = sel1, sel2, chains
        o[f"{key}_selfdist"], o[f"{key}_pbc"] = np.bool_(selfdist), np.bool_(pbc)
    elif name == "dist_trajectory_reduction":
        coords, box, g1, g2, ch1, ch2, selfdist, pbc, masses, r1, r2, results = a
        o[f"{key}_g1_atoms"], o[f"{key}_g1_offsets"] = _csr(g1)
        o[f"{key}_g2_atoms"], o[f"{key}_g2_offsets"] = _csr(g2)
        o[f"{key}_ch1"], o[f"{key}_ch2"] = ch1, ch2
        o[f"{key}_selfdist"], o[f"{key}_pbc"] = np.bool_(selfdist), np.bool_(pbc)
        o[f"{key}_r1"], o[f"{key}_r2"] = np.int32(r1), np.int32(r2)
        o[f"{key}_masses"] = np.asarray(masses, np.float32)
    else:
        assert name == "dist_trajectory_reduction_pairs", name
        coords, box, g1, g2, ch1, ch2, pbc, masses, r1, r2, results = a
        o[f"{key}_g1_atoms"], o[f"{key}_g1_offsets"] = _csr(g1)
        o[f"{key}_g2_atoms"], o[f"{key}_g2_offsets"] = _csr(g2)
        o[f"{key}_ch1"], o[f"{key}_ch2"] = ch1, ch2
        o[f"{key}_pbc"] = np.bool_(pbc)
        o[f"{key}_r1"], o[f"{key}_r2"] = np.int32(r1), np.int32(r2)
        o[f"{key}_masses"] = np.asarray(masses, np.float32)
    if coords_tag is None:
        o[f"{key}_coords"] = np.ascontiguousarray(coords, np.float32)
        o[f"{key}_box"] = np.ascontiguousarray(box, np.float32)
    else:                                           # the trajectory: read by the test from the committed XTC
        o[f"{key}_coords_tag"] = np.array(coords_tag)
        o[f"{key}_box_is_zero"] = np.bool_(not np.any(box))
    res = np.asarray(results, np.float32).copy()                    # what the compiled function wrote (before the drivers' post-processing)
    if coords_tag is not None:                                      # the trajectory's 200 frames: every fifth is kept
        o[f"{key}_frames"] = np.arange(0, res.shape[0], 5)
        res = res[::5].copy()
        if known is not None:
            o[f"{key}_known_frames"] = np.asarray(known, np.float64)[::5].copy()
    o[f"{key}_result"] = res
    if coords_tag is None and not np.array_equal(data, res):        # (pairs mode, truncate: the drivers post-process)
        o[f"{key}_projection"] = data
    if known is not None:
        known = np.asarray(known, np.float64)
        if coords_tag is None:
            o[f"{key}_known"] = known
        o[f"{key}_known_tol"], o[f"{key}_known_how"] = np.float64(tol), np.array(how)
        if how == "allclose":
            assert np.allclose(data, known, atol=tol, rtol=1e-5 if tol == 1e-8 else 0), (key, data, known)
        else:
            assert np.all(np.abs(data.ravel()[0] - known.ravel()[0]) < tol), (key, data, known)
    NAMES.append(key)
    o[f"{key}_truncate"] = np.float64(metr.truncate if getattr(metr, "truncate", None) is not None else -1.0)
    print(f"{key}: {name} -> result {o[f'{key}_result'].shape}, projection {data.shape}" + ("" if known is None else "  known answer holds"))
    return data


def main():
    for n in ("dist_trajectory", "dist_trajectory_reduction", "dist_trajectory_reduction_pairs"):
        _spy(n)
    # ---- test_metricdistance.py:99-181: analytic distances, wrapping, minimum over a group, pair order -------------------------
    mol = Molecule().empty(3)
    mol.name[:] = "C"; mol.element[:] = "C"
    mol.chain[:] = list(map(str, range(3)))
    mol.coords = np.zeros((3, 3, 2), dtype=np.float32)
    mol.coords[1, :, 0] = [3, 3, 3]; mol.coords[2, :, 0] = [5, 5, 5]
    mol.coords[1, :, 1] = [7, 7, 7]; mol.coords[2, :, 1] = [6, 6, 6]
    real = np.linalg.norm(mol.coords[[1, 2], :, :], axis=1).T
    record("trivial_open", MetricDistance("index 0", "index 1 2", metric="distances", periodic=None), mol, real, 1e-8)
    wrapped = np.linalg.norm(np.mod(mol.coords, 2)[[1, 2], :, :], axis=1).T
    mol.box = np.full((3, 2), 2, dtype=np.float32)
    record("trivial_wrapped", MetricDistance("index 0", "index 1 2", metric="distances", periodic="selections"), mol, wrapped, 1e-8)
    record("trivial_groupmin", MetricDistance("index 0", "index 1 2", metric="distances", periodic=None, groupsel1="all", groupsel2="all"), mol,
           np.min(real, axis=1)[:, None], 1e-8)
    mol = Molecule().empty(4)
    mol.name[:] = "C"; mol.element[:] = "C"
    mol.chain[:] = list(map(str, range(4)))
    mol.coords = np.zeros((4, 3, 2), dtype=np.float32)
    mol.coords[1, :, 0] = [1, 1, 1]; mol.coords[2, :, 0] = [3, 3, 3]; mol.coords[3, :, 0] = [5, 5, 5]
    mol.coords[1, :, 1] = [1, 1, 1]; mol.coords[2, :, 1] = [7, 7, 7]; mol.coords[3, :, 1] = [6, 6, 6]
    real = np.hstack((np.linalg.norm(mol.coords[[2, 3], :, :] - mol.coords[0], axis=1).T, np.linalg.norm(mol.coords[[2, 3], :, :] - mol.coords[1], axis=1).T))
    record("trivial_order", MetricDistance("index 0 1", "index 2 3", metric="distances", periodic=None), mol, real, 1e-8)
    # ---- :355-399: centre of mass / closest atom over selections and residues ---------------------------------------------------
    mol = Molecule().empty(4)
    mol.coords = np.array([[0, 0, 0], [-1, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)[:, :, None]
    mol.element[:] = "H"
    mol.resid[:] = [0, 1, 1, 0]
    fix = {"periodic": None, "groupsel1": "all", "groupsel2": "all", "groupreduce1": "com", "groupreduce2": "com"}
    record("com_a", MetricDistance("index 0", "index 1", **fix), mol, [[1.0]], 1e-5, "first")
    record("com_b", MetricDistance("index 0 1", "index 2", **fix), mol, [[1.5]], 1e-5, "first")
    record("com_c", MetricDistance("index 0 1 2", "index 3", **fix), mol, [[1.0]], 1e-5, "first")
    fix["groupsel1"] = "residue"
    record("com_residue", MetricDistance("index 0 1 2", "index 3", **fix), mol, [[1, 1]], 1e-8)
    fix["groupsel1"] = "all"; fix["groupreduce1"] = "closest"
    record("closest_all", MetricDistance("index 0 1 2", "index 3", **fix), mol, [[1]], 1e-8)
    fix["groupsel1"] = "residue"
    record("closest_residue", MetricDistance("index 0 1 2", "index 3", **fix), mol, [[1, 1.4142135]], 1e-8)
    # ---- :401-464: 3PTB, protein against benzamidine (the reference's test fetches 3PTB; the copy under tests/test_systemprepare is the
    # same entry -- the quoted numbers hold on it, asserted below) ------------------------------------------------------------------
    from moleculekit.periodictable import periodictable
    mol = Molecule(os.path.join(REF_TESTS, "test_systemprepare", "3PTB", "3PTB.pdb"))
    sel1, sel2 = "protein", "resname BEN"
    coms = []
    for s in (sel1, sel2):
        m = mol.copy(); m.filter(s)
        masses = np.array([periodictable[el].mass for el in m.element], dtype=np.float32)
        coms.append(np.sum(m.coords[:, :, 0] * masses[:, None], axis=0) / masses.sum())
    g = dict(groupsel1="all", groupsel2="all")
    record("ptb_com_com", MetricDistance(sel1, sel2, None, groupreduce1="com", groupreduce2="com", **g), mol, [[np.linalg.norm(coms[0] - coms[1])]], 1e-2, "first")
    record("ptb_com_closest", MetricDistance(sel1, sel2, None, groupreduce1="com", groupreduce2="closest", **g), mol, [[8.978174]], 1e-5, "first")
    record("ptb_closest_com", MetricDistance(sel1, sel2, None, groupreduce1="closest", groupreduce2="com", **g), mol, [[3.8286476]], 1e-5, "first")
    record("ptb_closest_closest", MetricDistance(sel1, sel2, None, groupreduce1="closest", groupreduce2="closest", **g), mol, [[2.8153415]], 1e-5, "first")
    # ---- :467-493: pairs mode ----------------------------------------------------------------------------------------------------
    s1 = np.array([0, 1, 2]).reshape(-1, 1); s2 = np.array([1, 2, 3]).reshape(-1, 1)
    ref = np.linalg.norm(mol.coords[s1.flatten(), :, 0] - mol.coords[s2.flatten(), :, 0], axis=1)
    record("ptb_pairs_atoms", MetricDistance(s1, s2, None, pairs=True), mol, ref[None, :], 1e-8)
    r1 = record("ptb_res_1_3", MetricDistance("residue 1", "residue 3", None), mol)
    r2 = record("ptb_res_2_4", MetricDistance("residue 2", "residue 4", None), mol)
    record("ptb_pairs_residues", MetricDistance("residue 1 2", "residue 3 4", None, pairs=True, groupsel1="residue", groupsel2="residue"), mol,
           [[r1.min(), r2.min()]], 1e-8)
    # ---- :329-352: the three meanings of `periodic` on the reference's trajectory -------------------------------------------------
    traj = os.path.join(REF_TESTS, "test_projections", "trajectory")
    mol = Molecule(os.path.join(traj, "filtered.pdb"))
    mol.read(os.path.join(traj, "traj.xtc"))
    a = "protein and resid 1 to 20 and noh"; b = "resname MOL and noh"
    d1 = record("periodic_selections", MetricDistance(a, b, periodic="selections"), mol, coords_tag="traj")
    d2 = record("periodic_chains", MetricDistance(a, b, periodic="chains"), mol, coords_tag="traj")
    assert np.allclose(d1, d2)
    tmp = mol.copy(); tmp.chain[:] = ""
    d3 = record("periodic_chains_one_chain", MetricDistance(a, b, periodic="chains"), tmp, coords_tag="traj")
    assert not np.allclose(d1, d3)
    # ---- :213-229: truncate (the drivers clip after the compiled call) --------------------------------------------------------------
    held = np.load(os.path.join(REF_TESTS, "test_projections", "metricdistance", "mindistances.npy"))
    record("mindistances_truncate", MetricDistance("protein and noh", b, periodic="selections", groupsel1="residue", groupsel2="all", truncate=3), mol,
           np.clip(held, 0, 3), 1e-3, coords_tag="traj")
    OUTD["names"] = np.array(NAMES)
    path = os.path.join(OUT, "metricdistance_known.npz")
    np.savez_compressed(path, **OUTD)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
