"""Shared by the CPU and GPU tiers: the KNOWN ANSWERS of the reference's own MetricDistance tests
(tests/golden/metricdistance_known.npz, made by tests/golden/make_golden_metricdistance_known.py from the scenarios of
/root/reference/tests/test_metricdistance.py:99-181, :213-229, :329-352, :355-464, :467-493).

`replay(fns, g, key, traj)` repeats the one call a scenario's projection makes into distance_utils -- the arguments as the
reference's drivers built them -- through `fns` (the oracle on the CPU tier, moleculekit_amd.distance_utils on the GPU tier);
`verify` holds the result against the compiled reference's (bit for bit) and against the number the reference's test asserts,
with that test's tolerance, after the drivers' post-processing (`truncate` clips, projections/util.py:86-87)."""
import numpy as np

from tests.cases import golden


def load():
    return golden("metricdistance_known.npz")


def _groups(atoms, offs):
    return [atoms[offs[i]:offs[i + 1]].tolist() for i in range(len(offs) - 1)]


def replay(fns, g, key, traj=None):
    """-> the [F, P] float32 array the compiled function writes.  `traj` = (coords, box) of the reference's trajectory."""
    if f"{key}_coords" in g.files:
        coords, box = g[f"{key}_coords"], g[f"{key}_box"]
    else:
        coords, box = traj
        if bool(g[f"{key}_box_is_zero"]):
            box = np.zeros_like(box)
    fn = str(g[f"{key}_fn"])
    pbc = bool(g[f"{key}_pbc"])
    F = coords.shape[2]
    if fn == "dist_trajectory":
        s1, s2, selfdist = g[f"{key}_sel1"], g[f"{key}_sel2"], bool(g[f"{key}_selfdist"])
        P = len(s1) * (len(s2) - 1) // 2 if selfdist else len(s1) * len(s2)
        res = np.full((F, P), -1.0, np.float32)
        fns.dist_trajectory(coords, box, s1, s2, g[f"{key}_chains"], selfdist, pbc, res)
        return res
    g1 = _groups(g[f"{key}_g1_atoms"], g[f"{key}_g1_offsets"]); g2 = _groups(g[f"{key}_g2_atoms"], g[f"{key}_g2_offsets"])
    r1, r2, masses = int(g[f"{key}_r1"]), int(g[f"{key}_r2"]), g[f"{key}_masses"]
    if fn == "dist_trajectory_reduction_pairs":
        res = np.full((F, len(g1)), -1.0, np.float32)
        fns.dist_trajectory_reduction_pairs(coords, box, g1, g2, g[f"{key}_ch1"], g[f"{key}_ch2"], pbc, masses, r1, r2, res)
        return res
    selfdist = bool(g[f"{key}_selfdist"])
    P = len(g1) * (len(g2) - 1) // 2 if selfdist else len(g1) * len(g2)
    res = np.full((F, P), -1.0, np.float32)
    fns.dist_trajectory_reduction(coords, box, g1, g2, g[f"{key}_ch1"], g[f"{key}_ch2"], selfdist, pbc, masses, r1, r2, res)
    return res


def verify(res, g, key):
    """-> True when the scenario carries a known answer of the reference's test (and it holds); asserts bit-exactness with the
    compiled reference either way."""
    frames = g[f"{key}_frames"] if f"{key}_frames" in g.files else None
    got = res if frames is None else res[frames]
    assert np.array_equal(got, g[f"{key}_result"]), f"{key}: not bit-exact with the compiled reference"
    if f"{key}_known_tol" not in g.files:
        return False
    proj = got.copy()
    trunc = float(g[f"{key}_truncate"])
    if trunc >= 0:
        proj[proj > trunc] = trunc
    known = g[f"{key}_known_frames"] if frames is not None else g[f"{key}_known"]
    tol = float(g[f"{key}_known_tol"])
    if str(g[f"{key}_known_how"]) == "first":                     # `abs(dist[0][0] - x) < tol`
        assert abs(float(proj[0, 0]) - float(known.ravel()[0])) < tol, (key, proj[0, 0], known)
    elif tol == 1e-8:                                             # plain np.allclose
        assert np.allclose(proj, known), (key, proj, known)
    else:
        assert np.allclose(proj, known, atol=tol), key
    return True
