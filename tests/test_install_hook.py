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
