"""CPU tier: host-side logic of the drop-in (grid / centre generation, radii, rotation, argument
handling) against golden outputs of the real reference; no GPU compute is called here."""
import json
import os

import numpy as np
import pytest

from moleculekit_amd import batch, distributed
from moleculekit_amd import voxeldescriptors as vd
from moleculekit_amd.util import boundingBox, rotationMatrix
from tests.cases import GOLDEN, golden


class Mol:
    """Duck-typed stand-in for moleculekit.molecule.Molecule (coords f32 [N,3,F], element, frame)."""

    def __init__(self, coords, element=None):
        self.coords = np.asarray(coords, np.float32)
        if self.coords.ndim == 2:
            self.coords = self.coords[:, :, None]
        self.element = element
        self.frame = 0

    def get(self, field, sel=None):
        assert field == "coords"
        return self.coords[:, :, self.frame].copy()


def test_getcenters_bbox_branch_bit_exact():
    g = golden("getcenters_cases.npz")
    for i in range(6):
        buf, vs = g[f"bbox{i}_params"]
        # the reference is called with python ints/floats; keep ints where the fixture used ints
        buf = int(buf) if float(buf).is_integer() else float(buf)
        vs = int(vs) if float(vs).is_integer() else float(vs)
        centers, nvox = vd.getCenters(Mol(g[f"bbox{i}_coords"]), buffer=buf, voxelsize=vs)
        assert np.array_equal(nvox, g[f"bbox{i}_nvoxels"])
        assert centers.dtype == np.float64 and centers.flags["C_CONTIGUOUS"]
        assert np.array_equal(centers, g[f"bbox{i}_centers"])


def test_getcenters_boxsize_branch_bit_exact():
    g = golden("getcenters_cases.npz")
    for i in range(5):
        vs = float(g[f"box{i}_voxelsize"])
        vs = int(vs) if vs.is_integer() else vs
        centers, nvox = vd.getCenters(None, boxsize=list(g[f"box{i}_boxsize"]), center=list(g[f"box{i}_center"]),
                                      voxelsize=vs)
        assert np.array_equal(nvox, g[f"box{i}_nvoxels"])
        assert np.array_equal(centers, g[f"box{i}_centers"])


@pytest.mark.parametrize("name", ["celecoxib", "ledipasvir"])
def test_getcenters_reference_held_fixtures(name):
    """centers and nvoxels must be EXACTLY equal (test_voxeldescriptors.py:52-53, :67-68)."""
    g = golden(f"{name}_ch7.npz")
    centers, nvox = vd.getCenters(Mol(g["coords"]), buffer=1)
    assert np.array_equal(centers, g["ref_centers"])
    assert np.array_equal(nvox, g["ref_nvoxels"])


def test_bounding_box_and_3ptb_grid():
    g = golden("3ptb_bbox_buffer8.npz")
    mol = Mol(g["coords"])
    bb = boundingBox(mol)
    assert bb.dtype == np.float32 and np.array_equal(bb, g["bbox"])
    centers, nvox = vd.getCenters(mol, buffer=8, voxelsize=1)
    assert np.array_equal(nvox, g["nvoxels"])
    assert np.array_equal(centers[:4], g["centers_first"]) and np.array_equal(centers[-4:], g["centers_last"])


def test_channel_radii():
    g = golden("3ptb_bbox_buffer8.npz")
    assert np.array_equal(vd._getChannelRadii(g["element"]), g["radii"])
    with open(os.path.join(GOLDEN, "vdw_radii.json")) as f:
        ref = json.load(f)
    from moleculekit_amd._vdw_radii import VDW_RADIUS
    assert VDW_RADIUS == ref and len(ref) == 118
    with pytest.raises(KeyError):
        vd._getChannelRadii(["Xx"])


def test_rotate_coordinates():
    g = golden("rotate_cases.npz")
    for r, exp in zip(g["rotations"], g["out"]):
        got = vd.rotateCoordinates(g["coords"], list(r), g["center"])
        assert np.allclose(got, exp, atol=1e-12)
    m = rotationMatrix([0, 0, 1], 1.5708)
    assert np.allclose(m.round(4), [[0, -1, 0], [1, 0, 0], [0, 0, 1]])


def test_grid_centers_cache_and_order():
    a = vd._getGridCenters(2, 3, 4, 0.5)
    assert a.shape == (2, 3, 4, 3) and a.dtype == np.float64
    flat = a.reshape(-1, 3)
    assert np.array_equal(flat[1], [0, 0, 0.5]) and np.array_equal(flat[4], [0, 0.5, 0])   # z fastest
    assert vd._getGridCenters(2, 3, 4, 0.5) is a                                          # lru_cache


def test_lattice_recognition():
    centers, nvox = vd.getCenters(None, boxsize=[6, 5, 4], center=[0.3, 1.0, -2.0], voxelsize=0.5)
    lat = vd._lattice_from_centers(centers)
    assert lat is not None
    bb, nv, vs = lat
    assert np.array_equal(nv, nvox) and vs == 0.5 and np.array_equal(bb, centers[0])
    assert vd._lattice_from_centers(centers[::-1]) is None
    jig = centers.copy(); jig[7, 1] += 1e-3
    assert vd._lattice_from_centers(jig) is None
    assert vd._lattice_from_centers(np.array([[0.0, 0, 0], [16, 24, -5]])) is None


def test_error_behaviour_matches_reference():
    coords = np.zeros((3, 3, 2), np.float32)
    ch = np.ones((3, 8))
    with pytest.raises(RuntimeError, match="single set of coordinates"):
        vd.getVoxelDescriptors(None, boxsize=[4, 4, 4], center=[0, 0, 0], usercoords=coords, userchannels=ch)
    with pytest.raises(RuntimeError, match="only support C implementation"):
        vd.getVoxelDescriptors(None, boxsize=[4, 4, 4], center=[0, 0, 0], usercoords=coords[:, :, 0],
                               userchannels=ch, method="cuda")


def test_calculate_occupancy_rejects_wrong_dtypes_like_cython():
    from moleculekit_amd.occupancy_utils import calculate_occupancy
    c, x, s, r = np.zeros((2, 3)), np.zeros((1, 3), np.float32), np.ones((1, 8)), np.zeros((2, 8))
    with pytest.raises(ValueError, match="dtype mismatch"):
        calculate_occupancy(c.astype(np.float32), x, s, r)
    with pytest.raises(ValueError, match="dtype mismatch"):
        calculate_occupancy(c, x.astype(np.float64), s, r)
    with pytest.raises(ValueError, match="dimensions"):
        calculate_occupancy(c, x, s[0], r)


def test_pack_items_and_max_images():
    coords, sig, offs = batch.pack_items([np.zeros((2, 3)), np.zeros((0, 3)), np.ones((5, 3))],
                                         [np.ones((2, 8)), np.ones((0, 8)), np.ones((5, 8))])
    assert coords.dtype == np.float32 and coords.shape == (7, 3) and list(offs) == [0, 2, 2, 7]
    assert batch.max_images_per_atom([[66.9] * 3], [48, 48, 48], 1.0) == 1
    assert batch.max_images_per_atom([[12.0, 12.0, 12.0]], [40, 40, 40], 1.0) == 5 ** 3
    with pytest.raises(ValueError):
        batch.max_images_per_atom([[9.0, 20, 20]], [8, 8, 8], 1.0)


def test_shard_bounds():
    assert list(distributed.shard_bounds(10, 4)) == [0, 3, 6, 8, 10]
    assert list(distributed.shard_bounds(3, 8)) == [0, 1, 2, 3, 3, 3, 3, 3, 3]
    b = distributed.shard_bounds(6, 2, weights=[100, 1, 1, 1, 1, 1])
    assert b[0] == 0 and b[-1] == 6 and b[1] == 1
    rng = np.random.default_rng(0)
    w = rng.integers(20, 51, size=1000)
    b = distributed.shard_bounds(1000, 8, weights=w)
    loads = [w[b[i]:b[i + 1]].sum() for i in range(8)]
    assert max(loads) - min(loads) <= 100 and np.all(np.diff(b) > 0)


def test_cube_export_is_byte_identical_to_the_reference(tmp_path):
    """`.cube` export of voxel grids (util.py:415-458); expected text written by the real reference."""
    from moleculekit_amd.util import readCube, writeCube, writeVoxelFeatures
    g = golden("cube_case.npz")
    fn = tmp_path / "a.cube"
    writeCube(g["arr"], str(fn), g["vecMin"], g["vecRes"])
    assert fn.read_text() == str(g["text"])
    back, meta = readCube(str(fn))
    assert np.array_equal(back, g["readback"]) and np.allclose(meta["org"], g["org"])
    # a grid whose value count is a multiple of six ends with a newline (like the reference's loop)
    writeCube(np.zeros((2, 3, 2)), str(fn), [0, 0, 0], [1, 1, 1])
    assert fn.read_text().endswith("0\n") and fn.read_text().count("\n") == 7 + 2
    # per-channel export of a getVoxelDescriptors-shaped result
    nv = np.array([3, 2, 4]); V = int(np.prod(nv))
    centers = (np.stack(np.meshgrid(np.arange(3), np.arange(2), np.arange(4), indexing="ij"), -1).reshape(V, 3) * 0.5 + [1.0, 2.0, 3.0])
    feats = np.random.default_rng(0).random((V, 8))
    files = writeVoxelFeatures(feats, centers, nv, str(tmp_path / "vox"))
    assert len(files) == 8 and files[7].endswith("vox_occupancies.cube")
    data, meta = readCube(files[2])
    assert np.allclose(data, feats[:, 2].reshape(3, 2, 4), rtol=1e-4) and data.shape == (3, 2, 4)
    assert np.allclose(meta["org"], (np.array([1.0, 2.0, 3.0]) - 0.25 + 0.25) / 0.52917725, atol=1e-5)


def test_lattice_recognition_follows_the_content_of_the_array():
    """`usercenters` callers pass the same array object call after call; what counts is what is IN it."""
    from moleculekit_amd.voxeldescriptors import _lattice_from_centers, getCenters
    c, nv = getCenters(boxsize=[6, 5, 4], center=np.array([1.0, 2.0, 3.0]), voxelsize=0.5)
    first = _lattice_from_centers(c)
    assert first is not None and np.array_equal(first[1], nv) and first[2] == 0.5
    first[0][:] = 99.0                                      # the caller may scribble on what it was handed
    assert np.array_equal(_lattice_from_centers(c)[0], c[0])
    c[7, 1] += 1e-3                                         # same object, different content
    assert _lattice_from_centers(c) is None
    c[7, 1] -= 1e-3
    assert _lattice_from_centers(c) is not None
    assert _lattice_from_centers(c[:, :2]) is None and _lattice_from_centers(c[:1]) is None


def test_native_lattice_recognition_agrees_with_its_numpy_specification():
    """mkamd_lattice_from_centers (what the drop-in asks of `usercenters`) against the numpy form it replaces: lattices
    of every shape class (one row, one plane, anisotropic counts, fractional voxel sizes, far origins), and non-lattices
    (a nudged centre, reversed order, a missing centre, noise at the tolerance, a NaN)."""
    from moleculekit_amd import voxeldescriptors as vd
    rng = np.random.default_rng(1)
    lattices = [vd.getCenters(boxsize=bs, center=np.array(ctr, float), voxelsize=vs)[0]
                for bs, vs, ctr in (([24, 24, 24], 1.0, [0, 0, 0]), ([6, 5, 4], 0.5, [1, 2, 3]), ([10, 1, 1], 1.0, [5, 5, 5]),
                                    ([1, 1, 7], 0.25, [0, 0, 0]), ([3, 4, 1], 1.0, [100, -50, 3]), ([2, 2, 2], 3.0, [1e4, 1e4, -1e4]))]
    others = []
    for c in lattices:
        j = c.copy(); j[len(j) // 2, 1] += 1e-6
        n = c.copy(); n[0, 0] = np.nan
        others += [j, c[::-1].copy(), c[:-1].copy(), c + rng.normal(0, 1e-12, c.shape), c + rng.normal(0, 3e-9, c.shape), n]
    n_lat = 0
    for c in lattices + others:
        a = vd._recognise_lattice(c)
        b = vd._recognise_lattice_numpy(np.asarray(c, dtype=np.float64))
        if np.isnan(c).any():
            assert a is None                                  # (the numpy form lets a NaN through its max(): not a lattice here)
            continue
        assert (a is None) == (b is None)
        if a is not None:
            n_lat += 1
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], np.asarray(b[1])) and a[2] == b[2]
    assert n_lat >= len(lattices)


def test_distance_wrappers_validate_what_the_library_reads_wholesale():
    """the C side copies 3 x F box floats and one chain id per atom; the reference reads the box only with
    pbc and a chain id only at the atoms it touches.  The wrappers reconcile the two BEFORE any pointer crosses."""
    from moleculekit_amd import distance_utils as du
    coords = np.zeros((6, 3, 4), np.float32)
    chains = np.zeros(6, np.uint32)
    box, ch = du._frame_inputs(coords, np.zeros((3, 1), np.float32), chains, False, (np.arange(6),))
    assert box.shape == (3, 4) and not box.any() and ch is chains
    with pytest.raises(ValueError, match=r"box must have shape \(3, 4\)"):
        du._frame_inputs(coords, np.zeros((3, 1), np.float32), chains, True, (np.arange(6),))
    _, ch = du._frame_inputs(coords, np.zeros((3, 4), np.float32), chains[:4], True, (np.array([0, 3], np.uint32),))
    assert ch.shape == (6,) and ch.dtype == np.uint32
    with pytest.raises(ValueError, match="atom 5 is used"):
        du._frame_inputs(coords, np.zeros((3, 4), np.float32), chains[:4], True, (np.array([5], np.uint32),))


def test_centres_cache_survives_concurrent_eviction():
    import threading
    errors = []

    def churn(seed):
        try:
            for i in range(200):
                vd._centersFromSpec(np.array([float(seed), float(i % 7), 0.0]), (3, 3, 3), 1.0)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=churn, args=(s,)) for s in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors and len(vd._CENTERS_CACHE) <= 4


def test_chunk_plan_of_the_streamed_drivers():
    """batch.chunk_plan: equal chunks, or a first chunk of `ramp` frames doubling up to `chunk`; the boundaries always cover the frames
    exactly once."""
    from moleculekit_amd.batch import chunk_plan
    assert chunk_plan(10, 4) == [0, 4, 8, 10] and chunk_plan(0, 4) == [0] and chunk_plan(3, 8) == [0, 3]
    assert chunk_plan(16384, 4096, 512) == [0, 512, 1536, 3584, 7680, 11776, 15872, 16384]
    assert chunk_plan(10, 4, 1) == [0, 1, 3, 7, 10] and chunk_plan(5, 8, 2) == [0, 2, 5]
    assert chunk_plan(10, 4, 4) == [0, 4, 8, 10] and chunk_plan(10, 4, 9) == [0, 4, 8, 10]       # a ramp that is no ramp
    for n, c, r in ((1000, 64, 8), (77, 13, 5), (5, 1, 3)):
        b = chunk_plan(n, c, r)
        assert b[0] == 0 and b[-1] == n and all(0 < y - x <= c for x, y in zip(b, b[1:]))
