"""CPU tier: the drop-in hook of SURVEY.md section 8b(3).  `install()` swaps `_getOccupancyC` of an installed
moleculekit (tools/voxeldescriptors.py:515-533, called at :356) for this package's; checked against a stub
`moleculekit.tools.voxeldescriptors` module shaped like the reference's (module-level `_getOccupancyC` looked up at
call time by `getVoxelDescriptors`).  The HIP library is not needed: the library call under the hook is recorded.
Also here: the default context is per host thread, and `getChannels` never silently changes backend."""
import sys
import gc
import threading
import weakref
import types

import numpy as np
import pytest


@pytest.fixture
def stub_moleculekit(monkeypatch):
    pkg, tools, vd = types.ModuleType("moleculekit"), types.ModuleType("moleculekit.tools"), types.ModuleType("moleculekit.tools.voxeldescriptors")

    def _getOccupancyC(coords, centers, channelsigmas):           # the reference's CPU path
        return np.full((centers.shape[0], channelsigmas.shape[1]), -1.0)

    def getVoxelDescriptors(coords, centers, channels):            # looks the helper up at call time, like :356
        return vd._getOccupancyC(coords, centers, channels)

    vd._getOccupancyC, vd.getVoxelDescriptors = _getOccupancyC, getVoxelDescriptors
    pkg.tools, tools.voxeldescriptors = tools, vd
    for name, mod in (("moleculekit", pkg), ("moleculekit.tools", tools), ("moleculekit.tools.voxeldescriptors", vd)):
        monkeypatch.setitem(sys.modules, name, mod)
    return vd


def test_install_swaps_and_uninstall_restores(stub_moleculekit, monkeypatch):
    import moleculekit_amd.voxeldescriptors as mine
    calls = []

    def fake_lattice(coords, offs, sigmas, origins, nvoxels, voxelsize, out=None, **kw):
        calls.append(("lattice", coords.dtype, sigmas.dtype, tuple(int(v) for v in nvoxels), float(voxelsize)))
        assert out is not None and out.dtype == np.float64          # the library widens while copying out
        out[...] = 0.0
        return out

    def fake_centers(centers, coords, sigmas, **kw):
        calls.append(("centers", centers.shape))
        return np.zeros((centers.shape[0], sigmas.shape[1]), np.float32)

    monkeypatch.setattr(mine._batch, "voxelize_lattice", fake_lattice)
    monkeypatch.setattr(mine._batch, "occupancy_centers", fake_centers)
    vd = stub_moleculekit
    original = vd._getOccupancyC
    assert mine.install() is original
    assert mine.install() is original                              # idempotent
    assert vd._getOccupancyC is not original and vd._getOccupancyC_reference is original

    coords = np.random.default_rng(0).normal(size=(5, 3))
    sig = np.ones((5, 8))
    centers, nvox = mine.getCenters(boxsize=[4, 3, 2], center=[0, 0, 0], buffer=0, voxelsize=1)
    out = vd.getVoxelDescriptors(coords, centers, sig)              # reference entry point -> our kernel path
    assert out.dtype == np.float64 and out.shape == (24, 8) and out.flags["C_CONTIGUOUS"]
    assert calls == [("lattice", np.dtype(np.float32), np.dtype(np.float64), (4, 3, 2), 1.0)]   # lattice recognised
    out = vd.getVoxelDescriptors(coords, centers[::-1].copy(), sig)  # arbitrary centres -> explicit-centre kernel
    assert calls[-1] == ("centers", (24, 3)) and out.shape == (24, 8)

    mine.uninstall()
    assert vd._getOccupancyC is original
    assert (vd.getVoxelDescriptors(coords, centers, sig) == -1.0).all()


def test_getchannels_backend_is_explicit(stub_moleculekit):
    import moleculekit_amd.voxeldescriptors as mine
    stub_moleculekit.getChannels = lambda *a: ("from-moleculekit", a[0])
    assert mine.CHANNELS_BACKEND == "table"
    assert mine.getChannels("mol", backend="moleculekit")[0] == "from-moleculekit"
    with pytest.raises(ValueError):
        mine.getChannels("mol", backend="nope")
    # the default never looks at the installed moleculekit: the table path gets the object (and rejects a non-molecule)
    with pytest.raises(Exception) as ei:
        mine.getChannels("mol")
    assert "from-moleculekit" not in str(ei.value)


def test_default_context_is_per_thread(monkeypatch):
    from moleculekit_amd import _lib

    class FakeCtx:
        made = 0

        def __init__(self, device):
            FakeCtx.made += 1
            self.device, self.closed = device, False

        def close(self):
            self.closed = True

    monkeypatch.setattr(_lib, "Context", FakeCtx)
    monkeypatch.setattr(_lib, "_tls", threading.local())
    main = _lib.default_context(0)
    assert _lib.default_context(0) is main and FakeCtx.made == 1
    other, kept_alive = [], []

    def worker(keep):
        c = _lib.default_context(0)
        assert _lib.default_context(0) is c
        other.append(weakref.ref(c))
        if keep:
            kept_alive.append(c)                 # handed to someone who outlives the thread

    t = threading.Thread(target=worker, args=(False,))
    t.start(); t.join()
    assert FakeCtx.made == 2 and _lib.default_context(0) is main
    t2 = threading.Thread(target=worker, args=(True,))
    t2.start(); t2.join()
    del t, t2
    gc.collect()
    # a context dies with its thread's storage -- unless someone else still holds it; nobody closes it from outside
    assert other[0]() is None
    assert other[1]() is kept_alive[0] and not kept_alive[0].closed and not main.closed


# ------------------------------------------------------------------------------------------------
# distance_utils row: install() also swaps the eight functions of moleculekit.distance_utils.
# The stub package below is shaped like the reference's callers: every one of them imports its function from
# `moleculekit.distance_utils` INSIDE the function body and calls it positionally (projections/util.py:22,100 behind
# MetricDistance; distance.py:242,278,308,360; molecule.py:3731).  The library under the hook is replaced by a recorder that
# answers with the oracle (test infrastructure), so the whole chain runs without a GPU.
# ------------------------------------------------------------------------------------------------
class _OracleCtx:
    """Stands in for _lib.Context: same host methods, answered by the oracle; records which entry points ran."""

    def __init__(self):
        self.calls = []

    def dist_trajectory_host(self, coords, box, sel1, sel2, chains, selfdist, pbc, squared, out):
        from oracle import oracle
        self.calls.append("dist_trajectory")
        out[...] = oracle.dist_trajectory(coords, box, sel1, sel2, chains, selfdist, pbc, squared=squared)

    def contacts_trajectory_host(self, coords, box, sel1, sel2, chains, selfdist, pbc, threshold):
        from oracle import oracle
        self.calls.append("contacts_trajectory")
        d2 = oracle.dist_trajectory(coords, box, sel1, sel2, chains, selfdist, pbc, squared=True)
        thr2 = np.float32(threshold) * np.float32(threshold)
        ii, jj = (np.triu_indices(len(sel1), 1) if selfdist else [a.ravel() for a in np.indices((len(sel1), len(sel2)))])
        offs, rows = [0], []
        for f in range(d2.shape[0]):
            hit = np.nonzero(d2[f] <= thr2)[0]
            rows.append(np.stack([sel1[ii[hit]], sel2[jj[hit]]], 1))
            offs.append(offs[-1] + len(hit))
        return np.asarray(offs, np.int64), np.concatenate(rows).astype(np.uint32).reshape(-1, 2)

    def dist_reduction_host(self, coords, box, g1a, g1o, g2a, g2o, ch1, ch2, selfdist, pairs, pbc, masses, r1, r2, out):
        from oracle import oracle
        self.calls.append("dist_reduction_pairs" if pairs else "dist_reduction")
        g1 = [g1a[g1o[i]:g1o[i + 1]].tolist() for i in range(len(g1o) - 1)]
        g2 = [g2a[g2o[i]:g2o[i + 1]].tolist() for i in range(len(g2o) - 1)]
        out[...] = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, selfdist, pbc, masses, r1, r2, pairs=pairs)

    def cdist_host(self, c1, c2, out):
        from oracle import oracle
        self.calls.append("cdist")
        out[...] = oracle.cdist(c1, c2)

    def pdist_host(self, c, out):
        from oracle import oracle
        self.calls.append("pdist")
        out[...] = oracle.pdist(c)


@pytest.fixture
def stub_distance_package(stub_moleculekit, monkeypatch):
    """moleculekit.distance_utils (the 'compiled' functions: they answer -1 / empty so that a call that still reaches them
    shows) + callers shaped like the reference's drivers."""
    import sys as _sys
    pkg = _sys.modules["moleculekit"]
    du = types.ModuleType("moleculekit.distance_utils")

    def _fill(results):
        results[...] = -1.0

    du.dist_trajectory = lambda coords, box, sel1, sel2, chains, selfdist, pbc, results: _fill(results)
    du.dist_trajectory_reduction = lambda c, b, g1, g2, c1, c2, selfdist, pbc, m, r1, r2, results: (_fill(results), results)[1]
    du.dist_trajectory_reduction_pairs = lambda c, b, g1, g2, c1, c2, pbc, m, r1, r2, results: (_fill(results), results)[1]
    du.contacts_trajectory = lambda c, b, s1, s2, ch, selfdist, pbc, thr=5: [[-1, -1] for _ in range(c.shape[2])]
    du.get_collisions = lambda c1, c2, thr: [-1, -1]
    du.cdist = lambda c1, c2, results: _fill(results)
    du.pdist = lambda c, results: _fill(results)
    du.squareform = lambda d: np.full((2, 2), -1.0, np.float32)

    putil, dist, molm = (types.ModuleType("moleculekit.projections.util"), types.ModuleType("moleculekit.distance"),
                         types.ModuleType("moleculekit.molecule"))

    def pp_calcDistances(mol, sel1, sel2, periodic):                       # like projections/util.py:12-85
        from moleculekit.distance_utils import dist_trajectory
        s1, s2 = np.where(sel1)[0].astype(np.uint32), np.where(sel2)[0].astype(np.uint32)
        chains = np.ones(mol.numAtoms, np.uint32); chains[s2] = 2
        res = np.zeros((mol.numFrames, len(s1) * len(s2)), np.float32)
        dist_trajectory(mol.coords, mol.box, s1, s2, chains, False, periodic is not None, res)
        return res

    def get_reduced_distances(mol, groups1, groups2, periodic, pairs=False):   # like projections/util.py:88-223
        from moleculekit.distance_utils import dist_trajectory_reduction, dist_trajectory_reduction_pairs
        ch1, ch2 = np.ones(len(groups1), np.uint32), np.full(len(groups2), 2, np.uint32)
        masses = np.ones(mol.numAtoms, np.float32)
        if pairs:
            res = np.zeros((mol.numFrames, len(groups1)), np.float32)
            dist_trajectory_reduction_pairs(mol.coords, mol.box, groups1, groups2, ch1, ch2, periodic is not None, masses, 0, 1, res)
        else:
            res = np.zeros((mol.numFrames, len(groups1) * len(groups2)), np.float32)
            dist_trajectory_reduction(mol.coords, mol.box, groups1, groups2, ch1, ch2, False, periodic is not None, masses, 0, 0, res)
        return res

    def cdist(a, b):                                                       # like distance.py:221-252
        from moleculekit.distance_utils import cdist
        res = np.zeros((a.shape[0], b.shape[0]), np.float32)
        cdist(a.astype(np.float32), b.astype(np.float32), res)
        return res

    def pdist(a):                                                          # like distance.py:255-282
        from moleculekit.distance_utils import pdist
        res = np.zeros(a.shape[0] * (a.shape[0] - 1) // 2, np.float32)
        pdist(a.astype(np.float32), res)
        return res

    def squareform(d):                                                     # like distance.py:285-305
        from moleculekit.distance_utils import squareform
        return np.array(squareform(d.astype(np.float32)))

    def calculate_contacts(mol, s1, s2, periodic, threshold=4):            # like distance.py:308-412
        from moleculekit.distance_utils import contacts_trajectory
        chains = np.zeros(mol.numAtoms, np.uint32)
        res = contacts_trajectory(mol.coords, mol.box, s1, s2, chains, False, periodic is not None, threshold)
        return [np.array(r, dtype=np.uint32).reshape(-1, 2) for r in res]

    def detect_collisions(c1, c2, thr):                                    # like molecule.py:3731
        from moleculekit.distance_utils import get_collisions
        return np.array(get_collisions(c1, c2, thr)).reshape(-1, 2)

    putil.pp_calcDistances, putil.get_reduced_distances = pp_calcDistances, get_reduced_distances
    dist.cdist, dist.pdist, dist.squareform, dist.calculate_contacts = cdist, pdist, squareform, calculate_contacts
    molm._detectCollisions = detect_collisions
    proj = types.ModuleType("moleculekit.projections")
    proj.util, pkg.distance_utils, pkg.projections, pkg.distance, pkg.molecule = putil, du, proj, dist, molm
    for name, mod in (("moleculekit.distance_utils", du), ("moleculekit.projections", proj), ("moleculekit.projections.util", putil),
                      ("moleculekit.distance", dist), ("moleculekit.molecule", molm)):
        monkeypatch.setitem(sys.modules, name, mod)
    return types.SimpleNamespace(du=du, putil=putil, dist=dist, mol=molm)


def test_install_swaps_distance_utils_and_reference_shaped_callers_reach_it(stub_distance_package, monkeypatch):
    import moleculekit_amd
    from moleculekit_amd import _lib, distance_utils as mine
    from oracle import oracle
    from tests.cases import golden

    ctx = _OracleCtx()
    monkeypatch.setattr(_lib, "default_context", lambda device=None: ctx)
    s = stub_distance_package
    originals = {n: getattr(s.du, n) for n in mine.HOOKED}
    g = golden("distance_cases.npz")
    mol = types.SimpleNamespace(coords=g["coords"], box=g["box"], numAtoms=g["coords"].shape[0], numFrames=g["coords"].shape[2])
    sel1 = np.zeros(mol.numAtoms, bool); sel1[g["sel1"]] = True
    sel2 = np.zeros(mol.numAtoms, bool); sel2[g["sel2"]] = True
    assert (s.putil.pp_calcDistances(mol, sel1, sel2, "selections") == -1).all()          # not installed yet

    moleculekit_amd.install()
    assert moleculekit_amd.install() is not None                                              # idempotent
    for n in mine.HOOKED:
        assert getattr(s.du, n) is getattr(mine, n), n
    chains = np.ones(mol.numAtoms, np.uint32); chains[g["sel2"]] = 2
    got = s.putil.pp_calcDistances(mol, sel1, sel2, "selections")
    assert np.array_equal(got, oracle.dist_trajectory(mol.coords, mol.box, g["sel1"], g["sel2"], chains, False, True))
    g1 = [list(map(int, x)) for x in g["groups1"]]; g2 = [list(map(int, x)) for x in g["groups2"]]
    got = s.putil.get_reduced_distances(mol, g1, g2, "selections")
    assert got.shape == (mol.numFrames, len(g1) * len(g2)) and (got >= 0).all()
    got = s.putil.get_reduced_distances(mol, g1[:5], g2[:5], None, pairs=True)
    assert got.shape == (mol.numFrames, 5) and (got >= 0).all()
    a, b = g["cdist_a3"], g["cdist_b3"]
    assert np.array_equal(s.dist.cdist(a, b), g["cdist_r3"]) and np.array_equal(s.dist.pdist(b), g["pdist_r3"])
    sq = s.dist.squareform(g["pdist_r3"])
    assert sq.shape == (b.shape[0], b.shape[0]) and np.array_equal(sq, sq.T) and sq[0, 1] == g["pdist_r3"][0]
    res = s.dist.calculate_contacts(mol, g["sel1"], g["sel2"], None, threshold=12.0)
    assert len(res) == mol.numFrames and all(r.ndim == 2 and r.shape[1] == 2 for r in res) and sum(len(r) for r in res) > 0
    col = s.mol._detectCollisions(np.ascontiguousarray(mol.coords[:20, :, 0]), np.ascontiguousarray(mol.coords[20:50, :, 0]), 14.0)
    assert np.array_equal(col.ravel(), g["collisions"])
    assert ctx.calls == ["dist_trajectory", "dist_reduction", "dist_reduction_pairs", "cdist", "pdist", "contacts_trajectory",
                         "contacts_trajectory"]

    moleculekit_amd.uninstall()
    for n in mine.HOOKED:
        assert getattr(s.du, n) is originals[n], n
    assert (s.putil.pp_calcDistances(mol, sel1, sel2, "selections") == -1).all()
    moleculekit_amd.install(distances=False)                                                   # the voxel helper alone
    assert s.du.dist_trajectory is originals["dist_trajectory"]
    moleculekit_amd.uninstall()


REF_BUILD = __import__("os").environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")


@pytest.mark.skipif(not __import__("os").path.exists(__import__("os").path.join(REF_BUILD, "moleculekit", "projections", "metricdistance.py"))
                    or not __import__("os").path.isdir("/root/reference/tests/test_projections/trajectory"),
                    reason="build container only: needs the real reference built in MOLECULEKIT_REF_BUILD")
def test_real_metricdistance_runs_on_the_installed_functions(monkeypatch):
    """With the REAL reference importable (the build container), install() makes its own MetricDistance drive this package's
    functions: the reference's test projections (tests/test_metricdistance.py:182-262) through the hook, the library answered
    by the oracle here, against the reference-held arrays at the reference's tolerance."""
    import os
    import subprocess
    code = r'''
import sys, os, types, numpy as np
sys.path.insert(0, %r); sys.path.insert(1, %r)
import moleculekit_amd
from moleculekit_amd import _lib
from tests.test_install_hook import _OracleCtx
from moleculekit.molecule import Molecule
from moleculekit.projections.metricdistance import MetricDistance, MetricSelfDistance
ctx = _OracleCtx(); _lib.default_context = lambda device=None: ctx
d = "/root/reference/tests/test_projections"
mol = Molecule(os.path.join(d, "trajectory", "filtered.pdb")); mol.read(os.path.join(d, "trajectory", "traj.xtc"))
# the reference's other callers of distance_utils, first on its own compiled functions ...
from moleculekit.distance import calculate_contacts, cdist, pdist, squareform
from moleculekit.molecule import _detectCollisions
few = mol.copy(); few.dropFrames(keep=[0, 7, 199])
s1, s2 = few.atomselect("protein and name CA"), few.atomselect("resname MOL and noh")
a, b = few.coords[s1, :, 0].copy(), few.coords[s2, :, 0].copy()
def others():
    con = calculate_contacts(few, s1, s2, "selections", threshold=30)
    return ([c.tolist() for c in con], cdist(a, b), pdist(b), squareform(pdist(b)), _detectCollisions(a, b, 40.0, np.arange(len(b))).tolist())
before = others()
moleculekit_amd.install()
# ... then through the hook: the same lists, the same bits
after = others()
assert before[0] == after[0] and before[4] == after[4] and sum(len(c) for c in before[0]) > 0 and len(before[4]) > 0
assert all(np.array_equal(x, y) for x, y in zip(before[1:4], after[1:4]))
assert ctx.calls == ["contacts_trajectory", "cdist", "pdist", "pdist", "contacts_trajectory"], ctx.calls
ctx.calls.clear()
r = MetricDistance("protein and name CA", "resname MOL and noh", metric="distances", periodic="selections").project(mol)
assert np.allclose(r, np.load(os.path.join(d, "metricdistance", "distances.npy")), atol=1e-3)
r = MetricDistance("protein and noh", "resname MOL and noh", periodic="selections", groupsel1="residue", groupsel2="all").project(mol)
assert np.allclose(r, np.load(os.path.join(d, "metricdistance", "mindistances.npy")), atol=1e-3)
r = MetricSelfDistance("protein and resid 1 to 50 and noh", groupsel="residue").project(mol)
assert np.allclose(r, np.load(os.path.join(d, "metricdistance", "selfmindistance.npy")), atol=1e-3)
assert ctx.calls == ["dist_trajectory", "dist_reduction", "dist_reduction"], ctx.calls
moleculekit_amd.uninstall()
print("HOOK-OK")
''' % (REF_BUILD, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "HOOK-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
