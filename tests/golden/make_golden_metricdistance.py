#!/usr/bin/env python3
"""Golden fixture for the distance_utils row on the reference's OWN real trajectory.

The reference's tests/test_metricdistance.py:182-279 project `tests/test_projections/trajectory/{filtered.pdb, traj.xtc}`
(4 507 atoms, 200 frames, periodic) with MetricDistance / MetricSelfDistance and compare with arrays the reference holds
(`tests/test_projections/metricdistance/*.npy`, atol 1e-3).  This script runs those projections with the REAL reference
built in a scratch directory (see make_golden.py) while recording, for every call the projections make into
`moleculekit.distance_utils`, the arguments (everything but the coordinates and the box, which come from the trajectory)
and the result.  Run in the build container only; nothing of the reference travels:

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_metricdistance.py

Stores DATA only:
  tests/golden/xtc/metricdistance_traj.xtc   the reference-held trajectory, byte for byte
  tests/golden/metricdistance_real.npz       per projection: selection / group index arrays, chain ids, flags, masses as the
                                             reference's drivers built them; the compiled reference's result (bit-exact pin);
                                             the reference-HELD arrays (the reference's own answers, atol 1e-3); checksums of
                                             the coordinates and the box the reference's reader decoded.
"""
import os
import shutil
import sys

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
REF_TESTS = os.environ.get("MOLECULEKIT_REF_TESTS", "/root/reference/tests")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)
import moleculekit.distance_utils as du  # noqa: E402
from moleculekit.molecule import Molecule  # noqa: E402
from moleculekit.projections.metricdistance import MetricDistance, MetricSelfDistance  # noqa: E402

CALLS = []


def _spy(name):
    real = getattr(du, name)

    def wrapper(*args):
        out = real(*args)
        CALLS.append((name, args))
        return out

    setattr(du, name, wrapper)        # the drivers import the function at call time (projections/util.py:22,100)


def _csr(groups):
    offs = np.zeros(len(groups) + 1, np.int64)
    offs[1:] = np.cumsum([len(g) for g in groups])
    return np.concatenate([np.asarray(g, np.int32) for g in groups]), offs


def main():
    for n in ("dist_trajectory", "dist_trajectory_reduction", "dist_trajectory_reduction_pairs"):
        _spy(n)
    traj = os.path.join(REF_TESTS, "test_projections", "trajectory")
    held = os.path.join(REF_TESTS, "test_projections", "metricdistance")
    mol = Molecule(os.path.join(traj, "filtered.pdb"))
    mol.read(os.path.join(traj, "traj.xtc"))
    os.makedirs(os.path.join(OUT, "xtc"), exist_ok=True)
    shutil.copyfile(os.path.join(traj, "traj.xtc"), os.path.join(OUT, "xtc", "metricdistance_traj.xtc"))
    os.chmod(os.path.join(OUT, "xtc", "metricdistance_traj.xtc"), 0o644)

    out = dict(natoms=np.int64(mol.numAtoms), nframes=np.int64(mol.numFrames),
               coords_bitsum=np.uint64(mol.coords.view(np.uint32).astype(np.uint64).sum()),
               coords_frame0=mol.coords[:, :, 0].copy(), coords_last=mol.coords[:, :, -1].copy(), box=mol.box.copy())

    projections = {
        # test_metricdistance.py:182-192
        "distances": (MetricDistance("protein and name CA", "resname MOL and noh", metric="distances", periodic="selections"), "distances.npy"),
        # :195-209
        "mindistances": (MetricDistance("protein and noh", "resname MOL and noh", periodic="selections", groupsel1="residue", groupsel2="all"),
                         "mindistances.npy"),
        # :230-262 (manual and automatic forms make the same call)
        "selfmindistance": (MetricSelfDistance("protein and resid 1 to 50 and noh", groupsel="residue"), "selfmindistance.npy"),
        # the centre-of-mass reductions of the same groups (no held array: pinned on the compiled reference only)
        "comdistances": (MetricDistance("protein and noh", "resname MOL and noh", periodic="selections", groupsel1="residue", groupsel2="all",
                                        groupreduce1="com", groupreduce2="com"), None),
        # :55-60 (every CA against every CA, non-periodic: the triangular pair list; four of its frames are kept below)
        "selfca": (MetricSelfDistance("protein and name CA", metric="distances"), None),
    }
    for key, (metr, heldfile) in projections.items():
        CALLS.clear()
        data = metr.project(mol)
        assert len(CALLS) == 1, (key, [c[0] for c in CALLS])
        name, a = CALLS[0]
        out[f"{key}_fn"] = np.array(name)
        if name == "dist_trajectory":
            coords, box, sel1, sel2, chains, selfdist, pbc, results = a
            out[f"{key}_sel1"], out[f"{key}_sel2"], out[f"{key}_chains"] = sel1, sel2, chains
            out[f"{key}_selfdist"], out[f"{key}_pbc"] = np.bool_(selfdist), np.bool_(pbc)
        else:
            assert name == "dist_trajectory_reduction"
            coords, box, g1, g2, ch1, ch2, selfdist, pbc, masses, r1, r2, results = a
            out[f"{key}_g1_atoms"], out[f"{key}_g1_offsets"] = _csr(g1)
            out[f"{key}_g2_atoms"], out[f"{key}_g2_offsets"] = _csr(g2)
            out[f"{key}_ch1"], out[f"{key}_ch2"] = ch1, ch2
            out[f"{key}_selfdist"], out[f"{key}_pbc"] = np.bool_(selfdist), np.bool_(pbc)
            out[f"{key}_r1"], out[f"{key}_r2"] = np.int32(r1), np.int32(r2)
            out["masses"] = np.asarray(masses, np.float32)
        assert np.array_equal(coords, mol.coords)
        out[f"{key}_box_is_zero"] = np.bool_(not np.any(box))
        res = np.asarray(data, np.float32)
        if key == "selfca":                       # 38 226 pairs x 200 frames would be 30 MB: four frames (of the test's skip=10 set)
            res = res[[0, 50, 100, 190]].copy()
            out["selfca_frames"] = np.array([0, 50, 100, 190], np.int64)
        out[f"{key}_result"] = res
        if heldfile:
            h = np.load(os.path.join(held, heldfile))
            out[f"{key}_held"] = h
            print(f"{key}: {name} -> {res.shape}; vs the held array: max |diff| {np.abs(res - h).max():.3g}, equal {np.array_equal(res, h)}")
            assert np.allclose(res, h, atol=1e-3)
        else:
            print(f"{key}: {name} -> {res.shape}")
    out["contacts_held"] = np.load(os.path.join(held, "contacts.npy"))       # held beside distances.npy (bool, threshold 8)
    np.savez_compressed(os.path.join(OUT, "metricdistance_real.npz"), **out)
    print("wrote", os.path.join(OUT, "metricdistance_real.npz"), os.path.getsize(os.path.join(OUT, "metricdistance_real.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
