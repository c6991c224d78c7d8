"""CPU tier: the library's HOST entry point mkamd_calculate_occupancy_cpu (csrc/cpu_occupancy.h) -- the product's
own double-precision implementation, explicit only -- against the outputs of the REAL reference (tests/golden/*.npz): the same
arithmetic per pair in range and an order-free maximum, so array_equal, not a tolerance."""
import numpy as np
import pytest

from moleculekit_amd import occupancy_utils
from moleculekit_amd.voxeldescriptors import getCenters, getVoxelDescriptors
from tests.cases import golden
from tests.synth import grid_origin


def _cpu(centers, coords, sigmas, n_threads=0, into=None):
    res = np.zeros((centers.shape[0], sigmas.shape[1])) if into is None else into
    occupancy_utils.calculate_occupancy_cpu(np.ascontiguousarray(centers, dtype=np.float64), np.ascontiguousarray(coords, dtype=np.float32),
                                            np.ascontiguousarray(sigmas, dtype=np.float64), res, n_threads=n_threads)
    return res


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_cfg1_3ptb_bit_exact(threads):
    g = golden("cfg1_3ptb.npz")
    assert np.array_equal(_cpu(g["centers"], g["coords"], g["sigmas"], threads), g["features"])


def test_dense_mixed_bit_exact():
    g = golden("dense_mixed.npz")
    o, nv = grid_origin(g["center"], g["boxsize"], float(g["voxelsize"]))
    from oracle import oracle          # (the checker only builds the lattice here)
    centers = oracle.grid_centers(o, nv, float(g["voxelsize"]))
    assert np.array_equal(_cpu(centers, g["coords"], g["sigmas"], 4), g["features"])


@pytest.mark.parametrize("C", [1, 3, 11])
def test_explicit_centres_any_channel_count(C):
    g = golden(f"explicit_C{C}.npz")
    assert np.array_equal(_cpu(g["centers"], g["coords"], g["sigmas"]), g["features"])


@pytest.mark.parametrize("name", ["cfg3_small.npz", "cfg5_small.npz"])
def test_small_molecule_batches(name):
    from oracle import oracle
    g = golden(name)
    vs = float(g["voxelsize"])
    for b in range(int(g["nmol"])):
        s, e = g["atom_offsets"][b], g["atom_offsets"][b + 1]
        o, nv = grid_origin(g["centers"][b], g["boxsize"], vs)
        assert np.array_equal(_cpu(oracle.grid_centers(o, nv, vs), g["coords"][s:e], g["sigmas"][s:e]), g["features"][b])


def test_in_place_maximum_and_edge_cases():
    """occupancy_utils.pyx:61: results = max(results, value) in place; an atom on a centre gives exactly 1; sigma 0 and NaN never
    store; NaN / inf coordinates and far outliers (the cell grid must not blow up) change nothing."""
    g = golden("explicit_C3.npz")
    start = np.full(g["features"].shape, 0.25)
    got = _cpu(g["centers"], g["coords"], g["sigmas"], into=start.copy())
    assert np.array_equal(got, np.maximum(g["features"], 0.25))
    coords = np.concatenate([g["coords"], np.array([[np.nan, 0, 0], [np.inf, 1, 1], [3e7, -2e7, 1e7], g["centers"][5]], dtype=np.float32)])
    sig = np.concatenate([g["sigmas"], np.array([[1.7, 1.7, 1.7], [1.7, 0, 1.7], [1.5, 1.5, 1.5], [0.0, np.nan, 1.2]])])
    got = _cpu(g["centers"], coords, sig, 2)
    want = g["features"].copy()
    from oracle import oracle
    want = oracle.calculate_occupancy(g["centers"], coords[[-1]], sig[[-1]], results=want)       # only the last atom is in range of anything
    assert np.array_equal(got, want) and not np.isnan(got).any()
    c32 = g["centers"][5].astype(np.float32).astype(np.float64)
    if np.array_equal(c32, g["centers"][5]):
        assert got[5, 2] == 1.0
    # no atoms / no centres / a NaN centre
    assert np.array_equal(_cpu(g["centers"], coords[:0], sig[:0]), np.zeros_like(g["features"]))
    assert _cpu(g["centers"][:0], coords, sig).shape == (0, 3)
    cen = g["centers"].copy(); cen[3] = np.nan
    got = _cpu(cen, g["coords"], g["sigmas"])
    assert np.array_equal(np.delete(got, 3, 0), np.delete(g["features"], 3, 0)) and np.all(got[3] == 0)


def test_contract_errors_are_the_reference_shaped_ones():
    g = golden("explicit_C1.npz")
    res = np.zeros_like(g["features"])
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        occupancy_utils.calculate_occupancy_cpu(g["centers"], g["coords"].astype(np.float64), g["sigmas"], res)
    with pytest.raises(ValueError, match="shape mismatch"):
        occupancy_utils.calculate_occupancy_cpu(g["centers"], g["coords"], g["sigmas"], res[:-1])


def test_method_cpu_is_the_reference_call_on_the_host():
    """getVoxelDescriptors(method="CPU"): the reference's call shape (voxeldescriptors.py:251-365) end to end without a device --
    centres, nvoxels and features of the cfg1 golden, bit for bit; `usercenters` comes back as the same object."""
    g = golden("cfg1_3ptb.npz")
    feats, centers, nvox = getVoxelDescriptors(None, boxsize=list(g["boxsize"]), center=g["center"], voxelsize=float(g["voxelsize"]),
                                               usercoords=g["coords"], userchannels=g["sigmas"], method="CPU")
    assert np.array_equal(centers, g["centers"]) and np.array_equal(nvox, g["nvoxels"]) and np.array_equal(feats, g["features"])
    assert feats.dtype == np.float64 and feats.flags["C_CONTIGUOUS"]
    uc = g["centers"][::7].copy()
    feats2, c2 = getVoxelDescriptors(None, usercenters=uc, usercoords=g["coords"][:, :, None], userchannels=g["sigmas"], method="cpu")
    assert c2 is uc and np.array_equal(feats2, g["features"][::7])
    with pytest.raises(RuntimeError, match="only support C implementation"):
        getVoxelDescriptors(None, usercenters=uc, usercoords=g["coords"], userchannels=g["sigmas"], method="numpy")


def test_random_configurations_against_the_checker():
    from oracle import oracle
    rng = np.random.default_rng(5)
    for _ in range(6):
        N, V, C = int(rng.integers(1, 400)), int(rng.integers(1, 3000)), int(rng.integers(1, 10))
        coords = (rng.normal(0, rng.uniform(2, 20), (N, 3))).astype(np.float32)
        centers = rng.normal(0, rng.uniform(2, 20), (V, 3))
        sig = rng.choice([0.0, 1.1, 1.7, 1.55, 2.27, 0.3], (N, C))
        assert np.array_equal(_cpu(centers, coords, sig, int(rng.integers(0, 5))), oracle.calculate_occupancy(centers, coords, sig))


def test_host_library_stands_alone_and_equals_the_copy_inside_libmkamd():
    """libmkamd_host.so (csrc/host_capi.cpp) is built by the plain C++ compiler and needs nothing of ROCm: no HIP runtime among
    its dependencies; the two entry points it exports give the bits of the copies inside libmkamd.so (clang, fp-contract off per
    function) -- and so the reference's."""
    import ctypes
    import subprocess
    from moleculekit_amd import _build, _lib
    path = _build.build_host()
    needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    assert "libamdhip64" not in needed and "libhsa" not in needed, needed
    g = golden("cfg1_3ptb.npz")
    cen, xyz, sig = (np.ascontiguousarray(g["centers"], np.float64), np.ascontiguousarray(g["coords"], np.float32),
                     np.ascontiguousarray(g["sigmas"], np.float64))
    a = np.zeros_like(g["features"]); b = np.zeros_like(g["features"])
    H, L = _lib.load_host(), _lib.load()
    args = lambda out: (cen.ctypes.data, cen.shape[0], xyz.ctypes.data, xyz.shape[0], sig.ctypes.data, sig.shape[1], out.ctypes.data)
    assert H.mkamd_calculate_occupancy_cpu(*args(a)) == 0 and L.mkamd_calculate_occupancy_cpu(*args(b)) == 0
    assert np.array_equal(a, b) and np.array_equal(a, g["features"])
    assert H.mkamd_calculate_occupancy_cpu_threads(None, 5, None, 5, None, 8, None, 1) == 1 and b"NULL" in H.mkamd_host_last_error()
    with pytest.raises(ValueError):
        occupancy_utils.calculate_occupancy_cpu(cen, xyz, sig.astype(np.float32), a)
