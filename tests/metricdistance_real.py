"""Shared by the CPU and GPU tiers: the reference's OWN MetricDistance projections on its real trajectory
(tests/golden/metricdistance_real.npz + tests/golden/xtc/metricdistance_traj.xtc, made by
tests/golden/make_golden_metricdistance.py from /root/reference/tests/test_metricdistance.py:182-262).

`run_projection(fns, traj, g, key)` repeats the one call a projection makes into distance_utils -- with the arguments
the reference's drivers built (stored) and the coordinates / box of the trajectory as THIS package's XTC reader decodes
them -- through `fns` (the oracle on the CPU tier, moleculekit_amd.distance_utils on the GPU tier)."""
import os

import numpy as np

from tests.cases import GOLDEN, golden

TRAJ = os.path.join(GOLDEN, "xtc", "metricdistance_traj.xtc")
KEYS = ("distances", "mindistances", "selfmindistance", "comdistances", "selfca")
HELD_ATOL = 1e-3            # what the reference's own tests assert (test_metricdistance.py:192,207,247)


def load():
    return golden("metricdistance_real.npz")


def read_trajectory(g):
    """coords [N,3,F] / box [3,F] in Angstrom from this package's host XTC reader; checked against what the reference's
    reader gave the reference's Molecule (bit patterns)."""
    from moleculekit_amd.xtc import XTCread
    t = XTCread(TRAJ)
    coords = np.ascontiguousarray(t.coords, np.float32)
    box = np.ascontiguousarray(t.box, np.float32)
    assert coords.shape == (int(g["natoms"]), 3, int(g["nframes"]))
    assert int(coords.view(np.uint32).astype(np.uint64).sum()) == int(g["coords_bitsum"])
    assert np.array_equal(coords[:, :, 0], g["coords_frame0"]) and np.array_equal(coords[:, :, -1], g["coords_last"])
    assert np.array_equal(box, g["box"])
    return coords, box


def _groups(atoms, offs):
    return [atoms[offs[i]:offs[i + 1]].tolist() for i in range(len(offs) - 1)]


def run_projection(fns, coords, box, g, key):
    pbc, selfdist = bool(g[f"{key}_pbc"]), bool(g[f"{key}_selfdist"])
    b = np.zeros_like(box) if bool(g[f"{key}_box_is_zero"]) else box       # periodic=None: the drivers pass a zero box
    if str(g[f"{key}_fn"]) == "dist_trajectory":
        s1, s2 = g[f"{key}_sel1"], g[f"{key}_sel2"]
        n1, n2 = len(s1), len(s2)
        P = n1 * (n2 - 1) // 2 if selfdist else n1 * n2
        res = np.zeros((coords.shape[2], P), np.float32)
        fns.dist_trajectory(coords, b, s1, s2, g[f"{key}_chains"], selfdist, pbc, res)
        return res
    g1 = _groups(g[f"{key}_g1_atoms"], g[f"{key}_g1_offsets"])
    g2 = _groups(g[f"{key}_g2_atoms"], g[f"{key}_g2_offsets"])
    P = len(g1) * (len(g2) - 1) // 2 if selfdist else len(g1) * len(g2)
    res = np.zeros((coords.shape[2], P), np.float32)
    fns.dist_trajectory_reduction(coords, b, g1, g2, g[f"{key}_ch1"], g[f"{key}_ch2"], selfdist, pbc, g["masses"],
                                  int(g[f"{key}_r1"]), int(g[f"{key}_r2"]), res)
    return res


def check(res, g, key):
    """-> (bit-exact with the compiled reference?, max |diff| to the reference-held array or None)"""
    live = g[f"{key}_result"]
    got = res[g["selfca_frames"]] if key == "selfca" else res
    exact = bool(np.array_equal(got, live))
    worst = None
    if f"{key}_held" in g.files:
        held = g[f"{key}_held"]
        assert np.allclose(res, held, atol=HELD_ATOL), f"{key}: off the reference-held array"
        worst = float(np.abs(res - held).max())
    return exact, worst
