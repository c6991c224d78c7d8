"""Parity cases shared by the CPU (emulated-kernel) tier and the GPU tier.

Each case is a dict: coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box (or None) and
``expected`` float64 [B,V,C] -- either straight from a golden fixture (outputs of the REAL
reference, tests/golden/make_golden.py) or computed by the oracle on the same seeded inputs.
"""
from __future__ import annotations

import os

import numpy as np

from oracle import oracle
from tests.synth import grid_origin, synth_config, synth_sigmas

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5   # BASELINE.json north_star: outputs within 1e-5 of the reference (float32 path)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def oracle_lattice(coords, atom_offsets, sigmas, origins, nvoxels, voxelsize, box=None):
    B = len(atom_offsets) - 1
    out = []
    for b in range(B):
        s, e = int(atom_offsets[b]), int(atom_offsets[b + 1])
        centers = oracle.grid_centers(origins[b], nvoxels, voxelsize)
        out.append(oracle.calculate_occupancy(centers, coords[s:e], sigmas[s:e],
                                              box=None if box is None else box[b]))
    return np.stack(out) if out else np.zeros((0, int(np.prod(nvoxels)), sigmas.shape[1]))


def _case(coords, offs, sigmas, origins, nvox, vs, box=None, expected=None):
    coords = np.ascontiguousarray(coords, np.float32).reshape(-1, 3)
    offs = np.asarray(offs, np.int64)
    origins = np.asarray(origins, np.float64).reshape(-1, 3)
    nvox = np.asarray(nvox, np.int64)
    if expected is None:
        expected = oracle_lattice(coords, offs, np.asarray(sigmas, np.float64), origins, nvox, vs, box)
    return dict(coords=coords, atom_offsets=offs, sigmas=sigmas, origins=origins, nvoxels=nvox,
                voxelsize=float(vs), box=box, expected=expected)


# ---- golden-backed cases (expected = real reference output) ---------------------------------------
def case_cfg1_3ptb():
    g = golden("cfg1_3ptb.npz")
    o, nv = grid_origin(g["center"], g["boxsize"], 1.0)
    assert np.array_equal(nv, g["nvoxels"])
    return _case(g["coords"], [0, len(g["coords"])], g["sigmas"], o[None], nv, 1.0,
                 expected=g["features"][None])


def case_dense_mixed():
    g = golden("dense_mixed.npz")
    o, nv = grid_origin(g["center"], g["boxsize"], float(g["voxelsize"]))
    return _case(g["coords"], [0, len(g["coords"])], g["sigmas"], o[None], nv, float(g["voxelsize"]),
                 expected=g["features"][None])


def case_cfg3_small():
    g = golden("cfg3_small.npz")
    o = np.stack([grid_origin(c, g["boxsize"], float(g["voxelsize"]))[0] for c in g["centers"]])
    return _case(g["coords"], g["atom_offsets"], g["sigmas"], o, g["nvoxels"], float(g["voxelsize"]),
                 expected=g["features"])


def case_cfg5_small():
    g = golden("cfg5_small.npz")
    o = np.stack([grid_origin(c, g["boxsize"], float(g["voxelsize"]))[0] for c in g["centers"]])
    return _case(g["coords"], g["atom_offsets"], g["sigmas"], o, g["nvoxels"], float(g["voxelsize"]),
                 expected=g["features"])


def case_pbc_small():
    g = golden("pbc_small.npz")
    o, nv = grid_origin(g["center"], g["boxsize"], 1.0)
    return _case(g["coords"], [0, len(g["coords"])], g["sigmas"], o[None], nv, 1.0,
                 box=g["box"][None].astype(np.float32), expected=g["features"][None])


def case_celecoxib_bbox():
    """Reference-held fixture: getCenters(mol, buffer=1) grid + channel 7 (test_voxeldescriptors.py:41-53)."""
    g = golden("celecoxib_ch7.npz")
    sig = np.zeros((len(g["coords"]), 8))
    sig[:, 7] = g["radii"] * (g["element"] != "H")
    exp = np.zeros((1, len(g["ref_centers"]), 8))
    exp[0, :, 7] = g["ref_features_ch7"]
    return _case(g["coords"], [0, len(g["coords"])], sig, g["ref_centers"][0][None], g["ref_nvoxels"], 1.0,
                 expected=exp)


# ---- oracle-backed cases (edge cases the fixtures do not cover) -------------------------------------
def case_ragged_batch():
    """Ragged batch incl. EMPTY items, a single-atom item, atoms far outside the grid, odd grid dims
    (partial tiles in every axis), float32 sigmas."""
    rng = np.random.default_rng(21)
    ns = [0, 1, 37, 0, 200, 5]
    coords = [rng.normal(0, 4, size=(n, 3)).astype(np.float32) for n in ns]
    coords[4][:20] += 100.0                      # far outside: must be dropped by the binning
    sig = np.concatenate([synth_sigmas(rng, n) for n in ns]).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(ns)])
    origins = rng.uniform(-9, -5, size=(len(ns), 3))
    return _case(np.concatenate(coords), offs, sig, origins, [13, 9, 21], 1.0)


def case_tiny_items():
    """Hundreds of tiny (and empty) items: one binning block spans far more cells than its LDS
    counters hold (-> it ranks with global atomics), the next ones straddle item boundaries;
    the last items are bigger, so that later blocks span few items (-> LDS ranking)."""
    rng = np.random.default_rng(27)
    ns = list(rng.integers(0, 13, size=420)) + [700, 900, 1500]
    coords = [rng.normal(0, 3.0, size=(n, 3)).astype(np.float32) for n in ns]
    sig = np.concatenate([synth_sigmas(rng, n) for n in ns])
    offs = np.concatenate([[0], np.cumsum(ns)])
    origins = rng.uniform(-7, -5, size=(len(ns), 3))
    return _case(np.concatenate(coords), offs, sig, origins, [12, 10, 9], 1.0)


def case_sorted_atoms():
    """Atom lists in spatially coherent order (runs of consecutive atoms in one cell, like residues / waters), with
    channel-less and out-of-grid atoms breaking the runs: the binning ranks such runs with one atomic each."""
    rng = np.random.default_rng(30)
    ns = [900, 5, 700]
    coords, sig = [], []
    for n in ns:
        c = rng.uniform(-9.0, 9.0, size=(n, 3)).astype(np.float32)
        key = np.floor((c + 16.0) / 8.0).astype(np.int64)
        c = c[np.lexsort((key[:, 2], key[:, 1], key[:, 0]))]
        sg = synth_sigmas(rng, n)
        sg[rng.random(n) < 0.15] = 0.0                       # dropped atoms inside the runs
        c[rng.random(n) < 0.05] += 80.0                      # atoms far outside the grid inside the runs
        coords.append(c); sig.append(sg)
    offs = np.concatenate([[0], np.cumsum(ns)])
    return _case(np.concatenate(coords), offs, np.concatenate(sig), np.tile([[-8.0, -8.0, -8.0]], (3, 1)), [16, 16, 16], 1.0)


def case_nonfinite_coords():
    """NaN / +-inf / huge coordinates: the reference's `dist2 < 25` is false for them, i.e. such atoms contribute
    nothing (and nothing may be indexed out of range on the way)."""
    rng = np.random.default_rng(29)
    n = 60
    c = rng.normal(0, 3.0, size=(n, 3)).astype(np.float32)
    c[3] = [np.nan, 0.0, 0.0]
    c[7] = [0.0, np.inf, 0.0]
    c[11] = [-np.inf, np.nan, 1.0]
    c[13] = [3.0e38, -3.0e38, 1.0e30]
    c[17] = [1.0e9, 0.0, 0.0]
    return _case(c, [0, n], synth_sigmas(rng, n), [[-6.0, -6.0, -6.0]], [12, 12, 12], 1.0)


def case_voxel07():
    """Non-dyadic voxel size (0.7 A) and a grid smaller than one tile."""
    rng = np.random.default_rng(22)
    n = 80
    c = rng.normal(0, 2.5, size=(n, 3)).astype(np.float32)
    return _case(c, [0, n], synth_sigmas(rng, n), [[-2.1, -1.4, -2.45]], [7, 5, 8], 0.7)


def case_voxel025():
    """Fine grid: 0.25 A voxels -> cutoff radius 20 voxels, 32-voxel cells."""
    rng = np.random.default_rng(23)
    n = 25
    c = rng.normal(0, 1.5, size=(n, 3)).astype(np.float32)
    return _case(c, [0, n], synth_sigmas(rng, n), [[-3.0, -3.0, -3.0]], [24, 24, 24], 0.25)


def case_voxel2():
    """Coarse grid: 2 A voxels: every radius is below 0.94 voxels, i.e. all classes take the exact form of the pair
    loop (kernels.h, fast_w_max)."""
    rng = np.random.default_rng(28)
    n = 1500
    c = rng.uniform(-16.0, 16.0, size=(n, 3)).astype(np.float32)
    return _case(c, [0, n], synth_sigmas(rng, n), [[-15.0, -17.0, -16.0]], [16, 18, 17], 2.0)


def case_voxel15():
    """1.5 A voxels: hydrogens (1.1 A = 0.73 voxels) take the exact form, the larger radii (>= 1.0 voxels) the one-fma
    form -- both in one call, dense enough that most voxels see both kinds."""
    rng = np.random.default_rng(29)
    n = 1200
    c = rng.uniform(-13.0, 13.0, size=(n, 3)).astype(np.float32)
    return _case(c, [0, n], synth_sigmas(rng, n), [[-12.0, -13.5, -12.75]], [17, 19, 18], 1.5)


def case_cutoff_exact(vs):
    """The strict cutoff d^2 < 25 on exactly representable geometry, with sigmas large enough that a wrongly included
    pair would show (sigma = 3 A at d = 5 A: 2.2e-3 against the tolerance of 1e-5): atoms ON voxel centres in every
    part of a tile (the one-fma form expands d^2 about the tile centre), so that voxels sit at exactly 5 A along the
    axes and on 3-4-0 triangles.  vs = 1 A, or 0.5 A (dyadic; distances in voxel units double)."""
    atoms = np.array([[0, 0, 0], [7, 7, 7], [3, 4, 4], [-8, 0, 7], [8, -8, -1]], np.float32)
    s = np.zeros((5, 8))
    s[:, 7] = 3.0
    s[0, 0] = 2.5
    s[2, 3] = 3.0
    s[3, 7] = 0.7 * vs                  # a class below 0.94 voxels: the exact form of the pair loop
    case = _case(atoms, [0, 5], s, [[-12.0, -12.0, -12.0]], [int(24 / vs) + 1] * 3, vs)
    # the oracle excludes the d = 5 voxels of the lone corner atom (channel 7: nothing else reaches them)
    n = int(24 / vs) + 1
    exp = case["expected"].reshape(n, n, n, 8)
    at = lambda x, y, z: exp[int((x + 12) / vs), int((y + 12) / vs), int((z + 12) / vs), 7]
    assert at(8, -8, 4) == 0.0 and at(8, -3, -1) == 0.0 and at(4, -5, -1) == 0.0 and at(8, -8, 3.0) > 1e-3
    return case


def _place_on_shell(rng, centre, delta_d2):
    """A float32 position whose squared distance to `centre` (double) is 25 + delta_d2 A^2 to within ~1e-9: a point on
    the 5 A sphere whose direction has one SMALL component (so that one float32 ulp along that axis moves d^2 by only
    ~1e-8), then an exhaustive search over the nearby float32 lattice points."""
    centre = np.asarray(centre, np.float64)
    k = int(rng.integers(0, 3))
    u = rng.normal(size=3)
    u[k] = 0.0
    u *= np.sqrt(1.0 - 0.003 ** 2) / np.linalg.norm(u)
    u[k] = 0.003 * rng.choice([-1.0, 1.0])
    base = (centre + np.sqrt(25.0 + delta_d2) * u).astype(np.float32)
    ulp = np.maximum(np.spacing(np.abs(base)).astype(np.float64), 1.2e-7)   # (a coordinate near 0 has tiny ulps: step coarser)
    steps = [np.arange(-3, 4), np.arange(-3, 4), np.arange(-3, 4)]
    steps[k] = np.arange(-3000, 3001)
    g = np.stack(np.meshgrid(*steps, indexing="ij"), axis=-1).reshape(-1, 3)
    cand = (base.astype(np.float64)[None, :] + g * ulp[None, :]).astype(np.float32)
    d = cand.astype(np.float64) - centre[None, :]
    err = np.abs((d * d).sum(axis=1) - 25.0 - delta_d2)
    i = int(err.argmin())
    return cand[i].copy(), float(err[i])


def case_cutoff_adversarial(vs, periodic=False):
    """The cut-off decision where float32 cannot make it: atoms with WIDE sigmas (Na 2.27 A, K 2.75 A, a user 3.0 A:
    the value at the cutoff is 7.7e-5 .. 2.2e-3 against the tolerance of 1e-5) placed so that one voxel centre of a
    NON-dyadic grid sits at d^2 = 25 + {0, +-1e-6, +-1e-5, +-1e-4} A^2 (d = 5 A +- 0, 1e-7, 1e-6, 1e-5) in double, the
    reference's arithmetic (occupancy_utils.pyx:53).  One item per (sigma, offset); every item also holds ordinary atoms
    in the same channel, so the exact re-evaluation has to see all of them.  `periodic`: the same through a box whose
    image, not the atom itself, carries the borderline distance."""
    rng = np.random.default_rng(77 + int(vs * 100) + (1000 if periodic else 0))
    n = int(round(16 / vs)) + 1
    origin = np.array([-8.13, -7.91, -8.37])
    coords, sig, offs, targets = [], [], [0], []
    for sigma in (2.27, 2.75, 3.0):
        for delta in (0.0, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4):
            iv = rng.integers(n // 2 - 2, n // 2 + 3, size=3)
            centre = oracle.grid_centers(origin, [n, n, n], vs).reshape(n, n, n, 3)[iv[0], iv[1], iv[2]]
            pos, err = _place_on_shell(rng, centre, delta)
            assert err < 5e-8, err
            others = (centre[None, :] + rng.normal(0, 1.0, size=(6, 3)) * 6.0).astype(np.float32)
            others = others[np.linalg.norm(others.astype(np.float64) - centre, axis=1) > 5.5]   # the target sees only the wide atom
            c = np.concatenate([pos[None, :], others])
            s = np.zeros((len(c), 8))
            s[0, 7] = sigma
            s[0, 2] = sigma
            s[1:, 7] = 1.7
            s[1:, 0] = 1.52
            if periodic:
                c[0] += np.array([20.0, -20.0, 40.0], np.float32) * 1.0      # only its image is in range
            coords.append(c); sig.append(s); offs.append(offs[-1] + len(c))
            targets.append(tuple(int(v) for v in iv))
    B = len(offs) - 1
    box = np.tile(np.array([[20.0, 20.0, 20.0]], np.float32), (B, 1)) if periodic else None
    case = _case(np.concatenate(coords), offs, np.concatenate(sig), np.tile(origin, (B, 1)), [n, n, n], vs, box=box)
    # the construction really straddles the cutoff: target voxels of the "inside" items are non-zero, "outside" ones zero
    exp = case["expected"].reshape(B, n, n, n, 8)
    k = 0
    for sigma in (2.27, 2.75, 3.0):
        for delta in (0.0, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4):
            v = exp[k][targets[k]][2]
            if not periodic and delta != 0.0:       # (delta = 0 lands within 1e-10 of the cutoff, on either side)
                assert (v > 5e-5) == (delta < 0), (sigma, delta, v)
            k += 1
    return case


def case_dense_with_wide_sigmas():
    """Dense tiles (900 atoms in a 7 A cube: more entries than any LDS tier holds, several rounds of the dense instance)
    AND wide sigmas on exactly representable geometry (atoms on voxel centres, sigma = 3 A in channel 7: voxels at exactly
    5 A along the axes and on 3-4-0 triangles, value at the cut-off 2.2e-3): the exact cut-off fix-up has to run AFTER
    the dense tiles of the same call -- one launch does both (k_tail), and its fix-up waves wait for its dense blocks."""
    rng = np.random.default_rng(77)
    n = 900
    c = rng.uniform(0, 7, size=(n, 3)).astype(np.float32)
    s = np.tile(rng.choice([1.1, 1.7, 1.52], size=(n, 1)), (1, 8)).astype(np.float64)
    s = np.where(rng.random((n, 8)) < np.asarray([1.0, 0.1, 0.1, 0.3, 1.0, 0.05, 0.0, 0.2])[None, :], s, 0.0)
    s[:, 7] = 0.0
    lattice = np.array([[0, 0, 0], [7, 7, 7], [3, 4, 4], [12, 3, 8], [-2, 9, 1]], np.float32)
    sl = np.zeros((len(lattice), 8))
    sl[:, 7] = 3.0
    return _case(np.concatenate([c, lattice]), [0, n + len(lattice)], np.concatenate([s, sl]), [[-4.0, -4.0, -4.0]], [24, 16, 16], 1.0)


def case_channels(C):
    """Channel counts other than 8 (channel groups: 1 -> padded group, 11 -> two groups)."""
    rng = np.random.default_rng(24 + C)
    n = 150
    c = rng.normal(0, 5, size=(n, 3)).astype(np.float32)
    s = rng.choice([0, 0, 1.1, 1.7, 1.52, 2.0], size=(n, C)).astype(np.float64)
    return _case(c, [0, n], s, [[-8.0, -8.0, -8.0]], [16, 17, 16], 1.0)


def case_special_sigmas():
    """sigma = 0 everywhere for some atoms, NaN, inf, negative, tiny and huge sigmas; an atom exactly
    ON a voxel centre (d = 0 -> value 1); atoms exactly 5 A from voxel centres (strict d^2 < 25)."""
    c = np.array([[0, 0, 0], [2.5, 2.5, 2.5], [-3, 1, 2], [4, -4, 0.5], [1, 1, 1], [-2, -2, -2],
                  [3, 0, 0], [0.25, 0.5, 0.75]], np.float32)
    s = np.zeros((8, 8))
    s[0, 7] = 1.7                # on a voxel centre; (5,0,0),(3,4,0) ... sit exactly at d = 5
    s[1, 0] = np.nan
    s[2, 1] = np.inf
    s[3, 2] = -1.52
    s[4, 3] = 1e-30
    s[5, 4] = 40.0
    s[6, 5] = 2.75
    # atom 7 keeps all-zero sigmas
    return _case(c, [0, 8], s, [[-8.0, -8.0, -8.0]], [17, 17, 17], 1.0)


def case_pbc_batch():
    """Periodic frames with per-frame boxes, box smaller than the grid along one axis (several
    images of one atom inside the grid) and unwrapped coordinates."""
    rng = np.random.default_rng(25)
    box = np.array([[18.0, 30.0, 26.0], [21.5, 24.0, 33.0]], np.float32)
    ns = [400, 350]
    coords, sig = [], []
    for b, n in enumerate(ns):
        x = rng.uniform(0, 1, size=(n, 3)) * box[b] + rng.integers(-2, 3, size=(n, 3)) * box[b]
        coords.append(x.astype(np.float32))
        sig.append(synth_sigmas(rng, n))
    offs = np.concatenate([[0], np.cumsum(ns)])
    origins = np.array([[-3.0, 2.0, 1.0], [0.5, -1.0, 4.0]])
    return _case(np.concatenate(coords), offs, np.concatenate(sig), origins, [24, 16, 20], 1.0, box=box)


def case_cfg4_small():
    """cfg4-shaped frames (BASELINE.json configs[3]) at reduced size: 3 frames x 3000 atoms, 31 A box."""
    rng = np.random.default_rng(26)
    L, n, F = 31.0, 3000, 3
    x = rng.uniform(0, L, size=(n, 3))
    frames = []
    for _ in range(F):
        frames.append(x.astype(np.float32))
        x = np.mod(x + rng.normal(0, 0.3, size=(n, 3)), L)
    s1 = synth_sigmas(rng, n)
    offs = np.arange(F + 1) * n
    origin = np.full(3, L / 2) - 12.0
    return _case(np.concatenate(frames), offs, np.concatenate([s1] * F), np.broadcast_to(origin, (F, 3)),
                 [24, 24, 24], 1.0, box=np.full((F, 3), L, np.float32))


LATTICE_CASES = {
    "cfg1_3ptb": case_cfg1_3ptb,
    "dense_mixed": case_dense_mixed,
    "cfg3_small": case_cfg3_small,
    "cfg5_small": case_cfg5_small,
    "pbc_small": case_pbc_small,
    "celecoxib_bbox": case_celecoxib_bbox,
    "ragged_batch": case_ragged_batch,
    "tiny_items": case_tiny_items,
    "nonfinite_coords": case_nonfinite_coords,
    "sorted_atoms": case_sorted_atoms,
    "voxel07": case_voxel07,
    "voxel025": case_voxel025,
    "voxel2": case_voxel2,
    "voxel15": case_voxel15,
    "cutoff_exact_1A": lambda: case_cutoff_exact(1.0),
    "cutoff_exact_05A": lambda: case_cutoff_exact(0.5),
    "cutoff_adversarial_1A": lambda: case_cutoff_adversarial(1.0),
    "cutoff_adversarial_07A": lambda: case_cutoff_adversarial(0.7),
    "cutoff_adversarial_pbc": lambda: case_cutoff_adversarial(1.0, periodic=True),
    "channels1": lambda: case_channels(1),
    "channels3": lambda: case_channels(3),
    "channels11": lambda: case_channels(11),
    "special_sigmas": case_special_sigmas,
    "dense_with_wide_sigmas": case_dense_with_wide_sigmas,
    "pbc_batch": case_pbc_batch,
    "cfg4_small": case_cfg4_small,
}


def check(case, got, tol=TOL):
    exp = case["expected"]
    got = np.asarray(got, np.float64).reshape(exp.shape)
    assert np.all(np.isfinite(got)), "non-finite output"
    err = np.abs(got - exp)
    assert err.max() <= tol, f"max-abs-err {err.max():.3e} > {tol:g} at {np.unravel_index(err.argmax(), err.shape)}"
    return float(err.max())
