"""CPU tier: the REAL kernel source (moleculekit_amd/csrc/kernels.h) and launch sequences
(pipeline.h) executed through the host SIMT emulation of tests/emu, checked against the oracle /
the golden reference outputs.  This validates the kernels' logic (tiling, binning, scan, periodic
images, channel groups, partial tiles) on a box without a GPU; the -m gpu tier repeats every case
through the C ABI on the MI355X."""
import numpy as np
import pytest

from oracle import oracle
from tests import emu_build as E
from tests.cases import LATTICE_CASES, TOL, check, golden, oracle_lattice


@pytest.mark.parametrize("name", sorted(LATTICE_CASES))
@pytest.mark.parametrize("tile_k", [8, 4])
def test_lattice_case(name, tile_k):
    """tile_k = 8 runs the one-launch per-item pre-pass (what small items get by default), tile_k = 4 the
    multi-kernel chain (what big items get): both pre-passes see every case."""
    if tile_k == 4 and name in ("cfg5_small", "pbc_small", "cfg4_small", "dense_mixed"):
        pytest.skip("covered with K=8 (emulation time)")
    case = LATTICE_CASES[name]()
    got, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"],
                                  case["nvoxels"], case["voxelsize"], box=case["box"], tile_k=tile_k,
                                  prepass_mode=1 if tile_k == 8 else 0)
    assert err == 0
    check(case, got)


@pytest.mark.parametrize("name", ["ragged_batch", "tiny_items", "special_sigmas", "pbc_batch", "channels11", "cfg1_3ptb"])
def test_per_item_and_chain_prepass_are_bit_identical(name):
    """The pre-pass only decides the ORDER of the records inside a cell and, per item instead of per call, the
    numbering of the sigma classes -- neither may change a single bit of the result (minima are order-free)."""
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    a, ea = E.voxelize_lattice(*args, box=case["box"], tile_k=8, prepass_mode=0)
    b, eb = E.voxelize_lattice(*args, box=case["box"], tile_k=8, prepass_mode=1)
    assert ea == 0 and eb == 0
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name,mode", [("ragged_batch", 0), ("cfg1_3ptb", 0), ("dense_mixed", 0), ("ragged_batch", 1), ("pbc_batch", 0)])
def test_counters_leave_every_call_zeroed(name, mode):
    """The cell counters and the dense-tile words are cleared by the kernels that read them last: a second and a third
    call on the same workspace run WITHOUT the memset (the emulation poisons fresh workspace, so a counter that was not
    put back would show) and return the first call's bits."""
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    fills1, fills3 = [0], [0]
    a, ea = E.voxelize_lattice(*args, box=case["box"], tile_k=8, prepass_mode=mode, fills=fills1)
    b, eb = E.voxelize_lattice(*args, box=case["box"], tile_k=8, prepass_mode=mode, repeat=3, fills=fills3)
    assert ea == 0 and eb == 0
    assert np.array_equal(a, b)
    assert fills3[0] == fills1[0]                       # the two extra calls issued no memset of their own
    check(case, b)


def test_per_item_class_tables_keep_mixed_batches_on_the_sorted_path():
    """40 distinct sigmas across the batch but <= 5 per item: the call-wide class table overflows (general path),
    per-item tables do not; one item with 20 distinct sigmas overflows on its own. Same values either way."""
    rng = np.random.default_rng(41)
    ns = [50, 60, 70, 80, 90, 100, 110, 120]
    coords = [rng.normal(0, 3.0, size=(n, 3)).astype(np.float32) for n in ns]
    radii = rng.uniform(0.9, 2.4, size=40)
    sig = []
    for i, n in enumerate(ns):
        pool = radii[5 * i:5 * i + 5] if i != 3 else radii[:20]
        sig.append(pool[rng.integers(0, len(pool), size=(n, 1))] * (rng.random((n, 8)) < 0.4))
    offs = np.concatenate([[0], np.cumsum(ns)])
    origins = np.tile(np.array([[-6.0, -6.0, -6.0]]), (len(ns), 1))
    args = (np.concatenate(coords), offs, np.concatenate(sig), origins, np.array([12, 12, 12]), 1.0)
    from tests.cases import oracle_lattice
    want = oracle_lattice(*args)
    a, _ = E.voxelize_lattice(*args, tile_k=8, prepass_mode=0)
    b, _ = E.voxelize_lattice(*args, tile_k=8, prepass_mode=1)
    assert np.abs(a - want).max() <= TOL
    assert np.array_equal(a, b)


def test_scan_kernels():
    rng = np.random.default_rng(0)
    for n in (0, 1, 63, 64, 255, 256, 4095, 4096, 4097, 3 * 4096, 70001):
        c = rng.integers(0, 9, size=n).astype(np.uint32)
        ref = np.concatenate([[0], np.cumsum(c)]).astype(np.uint32)
        assert np.array_equal(E.exclusive_scan(c), ref), n


@pytest.mark.parametrize("C", [1, 3, 11])
def test_explicit_centres_golden(C):
    g = golden(f"explicit_C{C}.npz")
    got = E.occupancy_centers(g["centers"], g["coords"], g["sigmas"])
    assert np.abs(got - g["features"]).max() <= TOL


def test_explicit_centres_pbc_and_f32_sigmas():
    rng = np.random.default_rng(3)
    c = rng.uniform(-20, 40, size=(500, 3)).astype(np.float32)
    s = rng.choice([0, 1.1, 1.7, 1.8], size=(500, 8)).astype(np.float32)
    centers = rng.uniform(0, 15, size=(300, 3))
    box = np.array([15.0, 17.0, 13.5])
    got = E.occupancy_centers(centers, c, s, box=box)
    exp = oracle.calculate_occupancy(centers, c, s.astype(np.float64), box=box)
    assert np.abs(got - exp).max() <= TOL


def test_grid_centers_bit_exact():
    g = golden("getcenters_cases.npz")
    for i in range(5):
        nv = g[f"box{i}_nvoxels"]
        bb = g[f"box{i}_center"] - g[f"box{i}_boxsize"] / 2
        assert np.array_equal(E.grid_centers(bb, nv, float(g[f"box{i}_voxelsize"])), g[f"box{i}_centers"])


def test_plan_choices():
    p = E.plan(1, 50000, 8, [64, 64, 64], 1.0)
    assert p["K"] == 4 and p["cs"] == 8 and p["h"] == 1 and p["ncx"] == 10      # single grid: more waves
    p = E.plan(64, 64 * 50000, 8, [64, 64, 64], 1.0)
    assert p["K"] == 8 and p["ntiles"] == 512
    p = E.plan(1000, 35000, 8, [24, 24, 24], 0.5)
    assert p["cs"] == 16 and p["rint"] == 10 and p["G"] == 1
    p = E.plan(4, 100, 11, [12, 24, 24], 1.0)
    assert p["K"] == 4 and p["G"] == 2                                          # x extent pads badly with K=8
    with pytest.raises(RuntimeError):
        E.plan(1, 10, 8, [24, 24, 24], 0.0)


def test_pbc_rejects_small_box():
    c = np.zeros((4, 3), np.float32)
    s = np.full((4, 8), 1.7)
    with pytest.raises(RuntimeError):
        E.voxelize_lattice(c, [0, 4], s, [[0, 0, 0]], [8, 8, 8], 1.0, box=np.array([[9.0, 20, 20]], np.float32))
    # device-side check (host does not know the boxes in the _dev entry point): flag is raised
    _, err = E.voxelize_lattice(c, [0, 4], s, [[0, 0, 0]], [8, 8, 8], 1.0,
                                box=np.array([[9.0, 20, 20]], np.float32), max_images=8)
    assert err & 2


def test_record_overflow_is_flagged():
    rng = np.random.default_rng(5)
    c = rng.uniform(0, 12, size=(50, 3)).astype(np.float32)
    s = np.full((50, 8), 1.7)
    # 12 A box inside a 40 A grid: several images per atom, but the caller claims only one
    _, err = E.voxelize_lattice(c, [0, 50], s, [[0, 0, 0]], [40, 40, 40], 1.0,
                                box=np.array([[12.0, 12.0, 12.0]], np.float32), max_images=1)
    assert err & 1


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "special_sigmas", "pbc_batch", "channels11", "nonfinite_coords", "voxel2", "voxel15"])
def test_sorted_and_general_paths_are_bit_identical(name):
    """The class-sorted tile path (cutoff, w and the plane constant c_k^2 hoisted out of the inner loop) must reproduce
    the general per-pair path bit for bit: min commutes exactly with monotone maps (adding a constant, clamping at 0,
    multiplying by w > 0); NaN / inf coordinates must be ignored by both."""
    case = LATTICE_CASES[name]()
    a, _ = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"],
                              case["nvoxels"], case["voxelsize"], box=case["box"], tile_k=8)
    b, _ = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"],
                              case["nvoxels"], case["voxelsize"], box=case["box"], tile_k=8, force_general=True)
    assert np.array_equal(a, b)


def test_more_than_16_sigma_classes_falls_back_to_general_path():
    rng = np.random.default_rng(31)
    n = 120
    c = rng.normal(0, 4, size=(n, 3)).astype(np.float32)
    s = np.where(rng.random((n, 8)) < 0.3, rng.uniform(0.8, 2.5, size=(n, 8)), 0.0)   # ~290 distinct sigmas
    case = dict(coords=c, atom_offsets=np.array([0, n]), sigmas=s, origins=np.array([[-8.0, -8, -8]]),
                nvoxels=np.array([16, 16, 16]), voxelsize=1.0, box=None)
    from tests.cases import oracle_lattice
    case["expected"] = oracle_lattice(c, case["atom_offsets"], s, case["origins"], case["nvoxels"], 1.0)
    got, err = E.voxelize_lattice(c, case["atom_offsets"], s, case["origins"], case["nvoxels"], 1.0)
    assert err == 0
    check(case, got)


def _dense_case(n, chan_prob, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 7, size=(n, 3)).astype(np.float32)          # n atoms in a 7 A cube
    s = np.tile(rng.choice([1.1, 1.7, 1.52], size=(n, 1)), (1, 8))
    s = np.where(rng.random((n, 8)) < np.asarray(chan_prob)[None, :], s, 0.0)
    case = dict(coords=c, atom_offsets=np.array([0, n]), sigmas=s, origins=np.array([[-4.0, -4, -4]]),
                nvoxels=np.array([24, 16, 16]), voxelsize=1.0, box=None)
    from tests.cases import oracle_lattice
    case["expected"] = oracle_lattice(c, case["atom_offsets"], s, case["origins"], case["nvoxels"], 1.0)
    return case


@pytest.mark.parametrize("n,chan_prob", [
    (900, [1.0] * 8),                                   # every channel alone exceeds the LDS arrays: all general
    (250, [1.0] * 8),                                   # ~250 entries per channel: several class-sorted rounds
    (900, [1.0, 0.1, 0.1, 0.3, 1.0, 0.05, 0.0, 0.2]),   # two oversized channels between channels that share rounds
])
def test_tile_denser_than_lds_capacity_runs_in_rounds(n, chan_prob):
    """More entries within reach of one tile than the LDS arrays hold: the class-sorted path takes the
    channels in rounds, a channel too dense on its own goes through the chunked path; other tiles are
    unaffected and everything stays bit-identical to the general path."""
    case = _dense_case(n, chan_prob, 32)
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], 1.0)
    got, err = E.voxelize_lattice(*args, tile_k=8)
    assert err == 0
    check(case, got)
    ref, _ = E.voxelize_lattice(*args, tile_k=8, force_general=True)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("tile_k", [4, 8])
def test_lds_tiers_are_bit_identical_and_report_overflow_statistics(tile_k):
    """The LDS tier (640/768/1024 entries per tile) only decides which tiles take the multi-round dense
    kernel; values do not depend on it. The statistics the dense kernel reports (tiles with more padded
    entries than each tier holds) are what the adaptive choice of the next call is based on."""
    case = _dense_case(110, [1.0] * 8, 33)           # ~880 entries near the cube: over tiers 0 and 1, under tier 2
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], 1.0)
    outs, fbs = [], []
    for tier in (0, 1, 2):
        fb = np.zeros(4, np.uint32)
        got, err = E.voxelize_lattice(*args, tile_k=tile_k, lds_tier=tier, feedback=fb)
        assert err == 0
        outs.append(got); fbs.append(fb.copy())
    check(case, outs[0])
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert all(np.array_equal(fbs[0], f) for f in fbs[1:])               # statistics do not depend on the tier run
    ntiles = int(np.prod(np.ceil(case["nvoxels"] / np.array([tile_k, 8, 8]))))
    assert fbs[0][3] == ntiles
    assert fbs[0][0] >= fbs[0][1] >= fbs[0][2] and fbs[0][0] > 0
    # adaptive: with these statistics as "previous call" the leanest tier with <= 5 % overflow is picked
    want = 0
    while want < 2 and fbs[0][want] * 20 > ntiles:
        want += 1
    assert E.choose_tier(-1, fbs[0]) == want
    assert E.choose_tier(1, fbs[0]) == 1
    assert E.choose_tier(-1, np.zeros(4, np.uint32)) == 0
    assert E.choose_tier(-1, np.array([100, 100, 100, 1000], np.uint32)) == 2
    assert E.choose_tier(-1, np.array([100, 10, 0, 1000], np.uint32)) == 1
    fb = fbs[0].copy()
    got, _ = E.voxelize_lattice(*args, tile_k=tile_k, lds_tier=-1, feedback=fb)
    assert np.array_equal(got, outs[0])


def test_fused_rotation_matches_rotate_then_voxelize():
    """rotateCoordinates (voxeldescriptors.py:78-114) fused into the binning stage == rotating on the
    host (golden-checked host function), casting to float32 (:519) and voxelizing."""
    from moleculekit_amd.batch import rotation_affines
    from moleculekit_amd.voxeldescriptors import rotateCoordinates
    case = LATTICE_CASES["cfg3_small"]()
    B = len(case["atom_offsets"]) - 1
    rng = np.random.default_rng(41)
    rots = rng.uniform(-np.pi, np.pi, size=(B, 3))
    cens = case["origins"] + 12.0
    fused, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                    case["voxelsize"], affine=rotation_affines(rots, cens))
    rc = case["coords"].copy()
    for b in range(B):
        s, e = case["atom_offsets"][b], case["atom_offsets"][b + 1]
        rc[s:e] = rotateCoordinates(case["coords"][s:e], list(rots[b]), cens[b]).astype(np.float32)
    plain, _ = E.voxelize_lattice(rc, case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    assert err == 0 and np.abs(fused - plain).max() <= 1e-5
    from tests.cases import oracle_lattice
    exp = oracle_lattice(rc, case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    assert np.abs(fused - exp).max() <= TOL
    assert np.abs(fused - case["expected"]).max() > 0.1          # the rotation really changed the grids


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "pbc_batch", "channels11", "cutoff_exact_1A", "voxel15", "special_sigmas"])
@pytest.mark.parametrize("tile_k", [8, 4])
def test_team_of_waves_per_tile_is_bit_identical(name, tile_k):
    """One grid per call (the reference's usage) runs a team of four waves per tile: shared candidate traversal, every
    fourth pair of a sub-bucket's entries per wave, the waves' minima merged through LDS before the epilogue.  Same
    arithmetic per (voxel, entry), minima are order-free and the class flush is monotone, so not a bit may differ from
    the one-wave kernel -- on the class-sorted path and, forced, on the general path."""
    if tile_k == 4 and name in ("pbc_batch", "voxel15"):
        pytest.skip("covered with K=8 (emulation time)")
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    one, e1 = E.voxelize_lattice(*args, box=case["box"], tile_k=tile_k, tile_team=0)
    team, e2 = E.voxelize_lattice(*args, box=case["box"], tile_k=tile_k, tile_team=1)
    assert e1 == 0 and e2 == 0
    assert np.array_equal(one, team)
    check(case, team)
    if name in ("ragged_batch", "channels11"):
        gen, e3 = E.voxelize_lattice(*args, box=case["box"], tile_k=tile_k, tile_team=1, force_general=True)
        assert e3 == 0 and np.array_equal(gen, one)
    if tile_k == 4 and name in ("cfg1_3ptb", "ragged_batch", "channels11", "special_sigmas"):
        # teams of 8 and 16 waves (a pocket's few dozen tiles): more waves than planes, so the epilogue splits the channels too
        for waves in (8, 16):
            big, e4 = E.voxelize_lattice(*args, box=case["box"], tile_k=4, tile_team=waves)
            assert e4 == 0 and np.array_equal(big, one), waves
        if name == "special_sigmas":
            gen, e5 = E.voxelize_lattice(*args, box=case["box"], tile_k=4, tile_team=16, force_general=True)
            assert e5 == 0 and np.array_equal(gen, one)


@pytest.mark.parametrize("name", ["cfg3_small", "cfg5_small", "tiny_items", "ragged_batch", "pbc_batch", "channels11", "special_sigmas",
                                  "cutoff_exact_1A", "voxel15", "cutoff_adversarial_1A"])
@pytest.mark.parametrize("prepass", [1, 0])
def test_workgroup_per_item_is_bit_identical(name, prepass):
    """Batches of ligand-sized items: one workgroup per item sorts the item's entries once for all its tiles
    (k_voxelize_items).  Same arithmetic per (voxel, entry) and the tile kernel's cull, so not a bit may differ from the
    wave-per-tile kernel -- small items on the sorted path, items of more than 384 entries (ragged_batch, cutoff cases
    with few atoms but many channels ...) and the forced general path on the unsorted one, on either pre-pass."""
    if prepass == 0 and name in ("cfg5_small", "pbc_batch", "voxel15", "cutoff_adversarial_1A"):
        pytest.skip("covered with the per-item pre-pass (emulation time)")
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    one, e1 = E.voxelize_lattice(*args, box=case["box"], tile_k=8, prepass_mode=prepass, tile_team=0, tile_items=0)
    itm, e2 = E.voxelize_lattice(*args, box=case["box"], tile_k=8, prepass_mode=prepass, tile_team=0, tile_items=1)
    assert e1 == 0 and e2 == 0
    assert np.array_equal(one, itm)
    check(case, itm)
    if name in ("tiny_items", "channels11"):
        gen, e3 = E.voxelize_lattice(*args, box=case["box"], tile_k=8, prepass_mode=prepass, tile_team=0, tile_items=1, force_general=True)
        assert e3 == 0 and np.array_equal(gen, one)
        k4, e4 = E.voxelize_lattice(*args, box=case["box"], tile_k=4, prepass_mode=prepass, tile_team=0, tile_items=1)
        k4r, _ = E.voxelize_lattice(*args, box=case["box"], tile_k=4, prepass_mode=prepass, tile_team=0, tile_items=0)
        assert e4 == 0 and np.array_equal(k4, k4r)


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "pbc_batch", "voxel07", "voxel025", "voxel2", "tiny_items", "sorted_atoms"])
def test_cell_size_only_moves_the_last_bits(name):
    """Half-cutoff cells (an A-B knob: fewer candidates per tile) against the cutoff-sized default: the cells decide which
    records a tile looks at before the exact cull, and the magnitude at which the cell-relative float32 offsets are
    rounded (smaller cells round finer) -- values agree to float32 noise, both within the parity bound."""
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    fine, e1 = E.voxelize_lattice(*args, box=case["box"], tile_k=8, fine_cells=True)
    coarse, e2 = E.voxelize_lattice(*args, box=case["box"], tile_k=8)
    assert e1 == 0 and e2 == 0
    assert np.abs(fine.astype(np.float64) - coarse).max() <= 5e-6
    check(case, fine)
    check(case, coarse)


def test_calculate_occupancy_returns_errors_instead_of_retrying_on_the_pairwise_kernel():
    """mkamd_calculate_occupancy (capi.hip) through the emulator: a getCenters lattice takes the tiled kernels, any other
    centre list the pairwise kernel, a lattice the plan refuses (more than 1023 cells per axis) falls through to the
    pairwise kernel -- and a FAILURE inside the lattice path (HIP error, allocation) comes back as the call's status
    with `results` untouched (round 2 retried it silently on the 100x slower kernel)."""
    from oracle import oracle
    rng = np.random.default_rng(5)
    coords = rng.uniform(0, 6, (40, 3)).astype(np.float32)
    sig = np.where(rng.random((40, 8)) < 0.3, 1.7, 0.0)
    centers = oracle.grid_centers(np.array([-1.0, -1.0, -1.0]), [8, 8, 8], 1.0)
    want = oracle.calculate_occupancy(centers, coords, sig)
    res = np.zeros_like(want)
    st, route = E.calculate_occupancy(centers, coords, sig, res)
    assert (st, route) == (0, 1) and np.abs(res - want).max() <= TOL
    jit = centers + rng.normal(0, 1e-3, centers.shape)                       # not a lattice
    res2 = np.full_like(want, 0.25)                                          # in-place max contract (occupancy_utils.pyx:61)
    st, route = E.calculate_occupancy(jit, coords, sig, res2)
    assert (st, route) == (0, 2) and np.abs(res2 - np.maximum(oracle.calculate_occupancy(jit, coords, sig), 0.25)).max() <= TOL
    for injected in (2, 6):                                                  # ST_EHIP, MKAMD_ENOMEM
        res3 = np.full_like(want, -7.0)
        st, route = E.calculate_occupancy(centers, coords, sig, res3, inject_lattice_status=injected)
        assert (st, route) == (injected, 0) and np.all(res3 == -7.0)
    st, route = E.calculate_occupancy(centers, coords, sig, np.zeros_like(want), inject_lattice_status=1)
    assert (st, route) == (0, 2)                                             # ST_EINVAL = the plan's refusal: pairwise serves it
    # a lattice the plan really refuses: 0.009 A voxels put the cutoff beyond 512 voxels
    tiny = oracle.grid_centers(np.array([2.0, 2.0, 2.0]), [4, 4, 4], 0.009)
    res4 = np.zeros((64, 8))
    st, route = E.calculate_occupancy(tiny, coords, sig, res4)
    assert (st, route) == (0, 2) and np.abs(res4 - oracle.calculate_occupancy(tiny, coords, sig)).max() <= TOL


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "pbc_batch", "cutoff_adversarial_1A", "special_sigmas", "voxel15"])
def test_tolerance_aware_reach_stays_inside_its_bound(name):
    """mkamd_ctx_set_value_tolerance (opt-in): every atom is culled per tile where it is worth less than eps.  The
    result may move by at most eps against the exact mode (a maximum over entries is 1-Lipschitz in each of them) and
    stays inside the 1e-5 parity bound against the reference; with eps = 0 nothing changes, bit for bit."""
    case = LATTICE_CASES[name]()
    kw = dict(box=case["box"]) if case["box"] is not None else {}
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    exact, _ = E.voxelize_lattice(*args, **kw)
    again, _ = E.voxelize_lattice(*args, value_tol=0.0, **kw)
    assert np.array_equal(exact, again)
    for eps in (1e-6, 5e-6):
        tol, err = E.voxelize_lattice(*args, value_tol=eps, **kw)
        assert err == 0
        assert np.abs(tol - exact).max() <= eps * 1.001                   # moved by no more than the tolerance
        assert np.all(tol <= exact)                                       # entries are only ever dropped
        assert np.abs(tol - case["expected"]).max() <= TOL                # and still inside the parity bound
    # the knob does something (3PTB has no hydrogens: only its N / O atoms drop to the 4.56 A level at eps = 5e-6)
    if name == "cfg1_3ptb":
        assert np.count_nonzero(tol != exact) > 0


def test_reach_levels_cover_the_radius_an_atom_needs():
    """Level l culls at reach^2 = (1 - 0.17 l) x 25 A^2; an atom gets the largest level whose reach still covers
    sigma x eps^(-1/12).  One atom per sigma on a fine grid: the voxels it loses are exactly those beyond ITS reach."""
    for sigma, eps in ((1.1, 1e-6), (1.52, 1e-6), (1.7, 1e-6), (0.8, 1e-6), (1.1, 5e-6)):
        need = min(5.0, sigma * eps ** (-1.0 / 12.0))
        coords = np.array([[12.03, 11.98, 12.01]], np.float32)
        sig = np.zeros((1, 8)); sig[0, 0] = sigma
        origin, nv = np.zeros((1, 3)), np.array([24, 24, 24])
        exact, _ = E.voxelize_lattice(coords, np.array([0, 1]), sig, origin, nv, 1.0)
        tol, _ = E.voxelize_lattice(coords, np.array([0, 1]), sig, origin, nv, 1.0, value_tol=eps)
        d = np.linalg.norm(oracle.grid_centers(origin[0], nv, 1.0) - coords[0].astype(np.float64), axis=1)
        lost = (tol[0, :, 0] == 0) & (exact[0, :, 0] > 0)
        assert np.abs(tol - exact).max() <= eps
        assert not lost[d < need - 1e-3].any()                            # nothing inside the radius the atom needs is lost


def _direct(case, repeat, **kw):
    words = np.zeros(4, np.uint32)
    kwargs = dict(box=case["box"]) if case["box"] is not None else {}
    out, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                  case["voxelsize"], direct=1, repeat=repeat, direct_words=words, prepass_mode=0, **kwargs, **kw)
    assert err == 0
    return out, words


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "voxel15", "cutoff_adversarial_1A", "special_sigmas", "channels11", "pbc_batch", "tiny_items"])
def test_direct_binning_is_bit_identical_with_the_chain_as_its_device_side_fallback(name):
    """k_bin_direct (round 3): records written in place at cell * capacity + rank with class ids from the table the
    previous call left; the count / scan / fill chain runs only when the pass gives up.  First call on a workspace: no
    table yet -> the pass fails, the chain does the work; second call: the pass succeeds.  Same bits every time."""
    case = LATTICE_CASES[name]()
    kwargs = dict(box=case["box"]) if case["box"] is not None else {}
    ref, _ = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                case["voxelsize"], prepass_mode=0, **kwargs)
    first, w1 = _direct(case, 1)
    assert np.array_equal(first, ref)
    second, w2 = _direct(case, 2)
    assert np.array_equal(second, ref)
    eligible = case["box"] is None and case["sigmas"].shape[1] <= 8
    if not eligible:                                   # periodic items / more than one channel group: the call has no direct pass
        assert w1[0] == 0xffffffff and w2[0] == 0xffffffff
        return
    assert w1[0] == 1                                  # no class table yet: gave up, the chain ran
    # atoms that carry several distinct sigmas (or NaN / inf ones) never go direct: the chain keeps serving such calls
    multi_sigma = any(len(set(r[r != 0])) > 1 for r in np.nan_to_num(case["sigmas"], nan=-1.0, posinf=-2.0))
    multi_sigma = multi_sigma or name == "special_sigmas"          # (a NaN sigma counts as "several": NaN != NaN)
    assert w2[0] == (1 if multi_sigma else 0), (name, w2)
    check(case, second)


def test_direct_binning_spills_full_cells_and_gives_up_when_the_spill_area_is_full():
    case = LATTICE_CASES["cfg1_3ptb"]()
    ref, _ = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                case["voxelsize"], prepass_mode=0)
    # four slots per cell: most of the 3PTB pocket's atoms go through the spill area, which every tile reads as one more run
    out, w = _direct(case, 2, cell_cap=4, spill_cap=4096)
    assert w[0] == 0 and w[1] > 500 and np.array_equal(out, ref)
    # ... and with a spill area of 64 slots the pass gives up and the chain does the call
    out, w = _direct(case, 2, cell_cap=4, spill_cap=64)
    assert w[0] == 1 and np.array_equal(out, ref)
    # the tolerance-aware reach levels ride in the direct records too
    tol_chain, _ = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                      case["voxelsize"], prepass_mode=0, value_tol=5e-6)
    tol_direct, w = _direct(case, 2, value_tol=5e-6)
    assert w[0] == 0 and np.array_equal(tol_direct, tol_chain)


def test_direct_binning_notices_a_sigma_the_old_table_lacks():
    """Calls on one workspace whose sigma set CHANGES: the table of the first batch does not cover the second one's
    classes -> the pass gives up, the chain leaves the new table, the next call goes direct again."""
    rng = np.random.default_rng(11)
    def batch(radii):
        coords = rng.uniform(0, 20, (300, 3)).astype(np.float32)
        sig = np.zeros((300, 8)); r = rng.choice(radii, 300)
        sig[:, 7] = r; sig[:, 0] = r * (rng.random(300) < 0.4)
        return coords, sig
    nv, org = np.array([20, 20, 20]), np.zeros((1, 3))
    words = np.zeros(4, np.uint32)
    lib = E.lib()
    import ctypes
    # two different batches through ONE backend need the C entry point's `repeat` to take different inputs: emulate by
    # running batch A twice (direct on the second call), then A's table against B in a fresh pair of calls
    a_c, a_s = batch([1.1, 1.7])
    b_c, b_s = batch([1.1, 1.7, 1.52, 2.27])
    ref_b, _ = E.voxelize_lattice(b_c, np.array([0, 300]), b_s, org, nv, 1.0, prepass_mode=0)
    out_a, _ = E.voxelize_lattice(a_c, np.array([0, 300]), a_s, org, nv, 1.0, direct=1, repeat=2, direct_words=words, prepass_mode=0)
    assert words[0] == 0
    out_b, _ = E.voxelize_lattice(b_c, np.array([0, 300]), b_s, org, nv, 1.0, direct=1, repeat=1, direct_words=words, prepass_mode=0)
    assert words[0] == 1 and np.array_equal(out_b, ref_b)                    # (a fresh backend: empty table, same code path as a stale one)


def test_direct_binning_feeds_the_exact_cutoff_fixup():
    """The fix-up waves of k_tail find an atom's sigma in the temp class descriptors: the direct pass must leave them as the
    chain does.  Wide single-sigma atoms (Na, user sigmas of 2.75 / 3 A) placed so that voxel centres sit at
    d^2 = 25 +- {0, 1e-6, 1e-5, 1e-4} A^2: without the fix-up several of these values are wrong by up to 2e-3."""
    from tests.cases import case_cutoff_adversarial
    case = case_cutoff_adversarial(1.0)
    sig = case["sigmas"].copy()
    widest = sig.max(axis=1, keepdims=True)
    sig = np.where(sig != 0, widest, 0.0)                                  # one sigma per atom: the direct pass can take them
    exp = oracle_lattice(case["coords"], case["atom_offsets"], sig, case["origins"], case["nvoxels"], case["voxelsize"])
    args = (case["coords"], case["atom_offsets"], sig, case["origins"], case["nvoxels"], case["voxelsize"])
    chain, _ = E.voxelize_lattice(*args, prepass_mode=0)
    words = np.zeros(4, np.uint32)
    direct, _ = E.voxelize_lattice(*args, prepass_mode=0, direct=1, repeat=2, direct_words=words)
    assert words[0] == 0                                                   # the second call did go direct
    assert np.abs(chain - exp).max() <= TOL and np.array_equal(direct, chain)


# ---- the one-launch pre-pass of small calls (k_bin_solo, round 3) ----------------------------------------------------
def _solo(case, repeat=1, **kw):
    words = np.zeros(4, np.uint32)
    kwargs = dict(box=case["box"]) if case["box"] is not None else {}
    fills = [0]
    out, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                  case["voxelsize"], direct=2, repeat=repeat, direct_words=words, fills=fills, **kwargs, **kw)
    assert err == 0
    return out, words, fills[0]


def _chain(case, **kw):
    kwargs = dict(box=case["box"]) if case["box"] is not None else {}
    out, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                  case["voxelsize"], direct=0, prepass_mode=0, **kwargs, **kw)
    assert err == 0
    return out


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "voxel15", "cutoff_adversarial_1A", "special_sigmas", "dense_mixed",
                                  "channels11", "pbc_batch", "tiny_items", "nonfinite_coords", "dense_with_wide_sigmas", "cfg3_small", "voxel025"])
def test_solo_prepass_is_bit_identical_with_the_chain(name):
    """k_bin_solo: the whole pre-pass of a small call in one launch -- records in the direct layout, the class table kept
    across calls and extended on the spot, every atom kind handled (several sigmas per atom, NaN / inf sigmas, more classes
    than ids), nothing enqueued behind it.  First call on a workspace (empty table: every class is inserted) and third call
    (table there, counters zeroed by k_tail) against the count / scan / fill chain: the same bits."""
    if name not in LATTICE_CASES:
        pytest.skip("no such case")
    case = LATTICE_CASES[name]()
    ref = _chain(case)
    three, w3, f3 = _solo(case, 3)
    assert np.array_equal(three, ref)
    eligible = case["box"] is None and case["sigmas"].shape[1] <= 8 and len(case["coords"]) > 0
    if eligible:
        assert w3[0] == 0                              # the pass never gives up ...
        assert w3[1] == 0                              # ... and k_tail left the control words (and the counters) zero
    if name == "cfg1_3ptb":                            # the FIRST call on a workspace alone (every class is inserted)
        one, w1, f1 = _solo(case, 1)
        assert np.array_equal(one, ref) and w1[0] == 0
        assert f3 == f1                                # no memset per call: the later calls found everything clean
    check(case, three)


def test_solo_prepass_with_more_sigma_classes_than_ids():
    """20 distinct radii: the table fills up, the overflow word is raised, the tile kernels read w from the records
    (general path); k_tail empties the table, so the second call on the workspace goes the same way."""
    rng = np.random.default_rng(5)
    n = 400
    coords = rng.uniform(-2, 22, (n, 3)).astype(np.float32)
    radii = np.linspace(1.0, 2.9, 20)
    sig = np.zeros((n, 8))
    r = rng.choice(radii, n)
    sig[:, 7] = r
    sig[:, 2] = r * (rng.random(n) < 0.5)
    sig[:, 4] = rng.choice(radii, n) * (rng.random(n) < 0.2)      # (some atoms carry two different radii)
    case = dict(coords=coords, atom_offsets=np.array([0, n]), sigmas=sig, origins=np.zeros((1, 3)), nvoxels=np.array([20, 20, 20]),
                voxelsize=1.0, box=None)
    exp = oracle_lattice(coords, case["atom_offsets"], sig, case["origins"], case["nvoxels"], 1.0)
    ref = _chain(case)
    for rep in (1, 2):
        out, w, _ = _solo(case, rep)
        assert w[0] == 0 and np.array_equal(out, ref)
    assert np.abs(ref - exp).max() <= TOL


def test_solo_prepass_spills_full_cells():
    case = LATTICE_CASES["cfg1_3ptb"]()
    ref = _chain(case)
    out, w, _ = _solo(case, 2, cell_cap=4)             # four slots per cell: most atoms go through the spill area
    assert w[1] == 0 and np.array_equal(out, ref)
    tol_chain = _chain(case, value_tol=5e-6)           # the tolerance-aware reach levels ride in these records too
    tol_solo, _, _ = _solo(case, 1, value_tol=5e-6)
    assert np.array_equal(tol_solo, tol_chain)


@pytest.mark.parametrize("mode", [1, 2])
def test_direct_layouts_keep_the_spilled_records_of_different_items_apart(mode):
    """Several items whose cells overflow (two slots per cell): a spilled record carries no item, so every item has a spill
    area of its own -- one shared area made the tiles of an item see the other items' atoms."""
    case = LATTICE_CASES["ragged_batch"]()
    ref = _chain(case)
    words = np.zeros(4, np.uint32)
    out, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                  case["voxelsize"], direct=mode, repeat=2, direct_words=words, cell_cap=2, spill_cap=4096,
                                  **({"prepass_mode": 0} if mode == 1 else {}))
    assert err == 0 and words[0] == 0 and np.array_equal(out, ref)


def test_solo_prepass_feeds_the_exact_cutoff_fixup():
    from tests.cases import case_cutoff_adversarial
    case = case_cutoff_adversarial(1.0)
    exp = oracle_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    ref = _chain(case)
    out, w, _ = _solo(case, 2)
    assert np.array_equal(out, ref) and np.abs(out - exp).max() <= TOL


def test_small_calls_take_the_solo_prepass_by_default():
    """Automatic mode: a one-molecule call (team regime) is binned by k_bin_solo -- seen through the direct control words,
    which only exist when the call had a direct pass."""
    case = LATTICE_CASES["cfg1_3ptb"]()
    words = np.zeros(4, np.uint32)
    out, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                  case["voxelsize"], direct_words=words)
    assert err == 0 and words[0] == 0 and np.array_equal(out, _chain(case))
    out, err = E.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                  case["voxelsize"], direct_words=words, prepass_mode=1)
    assert words[0] == 0xffffffff and np.array_equal(out, _chain(case))        # the caller chose a pre-pass: no direct pass


def _frames_of_one_molecule(seed, n, F, C=8, pbc=False, wide=False, multi=False):
    """F sets of coordinates of one n-atom molecule (a jittering trajectory) + its sigma matrix."""
    from tests.synth import synth_sigmas
    rng = np.random.default_rng(seed)
    sig = synth_sigmas(rng, n).astype(np.float64)
    if C != 8:
        sig = np.concatenate([sig, sig[:, : C - 8] * 0.9], axis=1) if C > 8 else sig[:, :C]
    if wide:
        sig[::7, 0] = 2.27            # Na: wide enough for the exact cut-off fix-up
    if multi:
        sig[::5, 1] = 1.3             # atoms with several distinct sigmas
        sig[::5, 3] = 1.9
    base = rng.uniform(-9.0, 9.0, size=(n, 3))
    frames = np.stack([base + rng.normal(0, 0.4, size=(n, 3)) for _ in range(F)]).astype(np.float32)
    box = np.tile(np.array([[21.0, 23.0, 22.0]], np.float32), (F, 1)) if pbc else None
    return frames.reshape(F * n, 3), sig, box


@pytest.mark.parametrize("kind", ["plain", "pbc", "wide", "multi", "channels11", "f32"])
@pytest.mark.parametrize("tile_k", [8, 4])
def test_topology_calls_are_bit_identical_with_plain_calls(kind, tile_k):
    """Round 5: frames of ONE molecule voxelized with its topology handle (class ids, channel words, class table
    and the wide flag built once by k_topology_classes / k_topology_ids; binning by k_bin_count<.., TOPO> / k_bin_fill<.., TOPO>)
    against the plain call with the sigma matrix repeated per frame: the same records, the same features, bit for bit -- open
    and periodic frames, wide sigmas (the exact fix-up reads the molecule's sigmas by the index inside the item), atoms with
    several sigmas, two channel groups, float32 sigmas; repeated calls on one workspace."""
    if tile_k == 4 and kind in ("multi", "channels11", "f32"):
        pytest.skip("covered with K=8 (emulation time)")
    n, F = 230, 3
    coords, sig, box = _frames_of_one_molecule(11, n, F, C=11 if kind == "channels11" else 8, pbc=kind == "pbc", wide=kind == "wide",
                                               multi=kind == "multi")
    if kind == "f32":
        sig = sig.astype(np.float32)
    origins = np.tile([[-10.0, -11.0, -9.5]], (F, 1))
    nv = [20, 22, 19]
    offs = np.arange(F + 1) * n
    plain, e0 = E.voxelize_lattice(coords, offs, np.tile(sig, (F, 1)), origins, nv, 1.0, box=box, tile_k=tile_k, prepass_mode=0)
    topo, e1, wide = E.voxelize_lattice_topo(coords, sig, F, origins, nv, 1.0, box=box, tile_k=tile_k, repeat=2)
    assert e0 == 0 and e1 == 0 and wide == (kind in ("wide", "multi"))        # (1.9 A and 2.27 A are wider than 1.81 A)
    assert np.array_equal(plain, topo)
    if kind == "wide":                       # and the fix-up really ran on this case: values differ from a run without the wide atoms' exactness
        assert np.abs(plain.astype(np.float64) - oracle_lattice(coords, offs, np.tile(sig, (F, 1)), origins, np.array(nv), 1.0, box)).max() <= TOL


def test_topology_call_with_an_item_of_another_length_is_flagged():
    """every item must be the topology's atom count long: checked on the device (MK_ERR_TOPOLOGY = 8), not assumed"""
    n, F = 100, 3
    coords, sig, _ = _frames_of_one_molecule(3, n, F)
    offs = np.array([0, n, 2 * n - 10, 3 * n])          # the right total, the wrong split
    origins = np.tile([[-10.0, -10.0, -10.0]], (F, 1))
    _, err, _ = E.voxelize_lattice_topo(coords, sig, F, origins, [20, 20, 20], 1.0, atom_offsets=offs)
    assert err & 8


def test_periodic_image_ranges_without_divisions_are_the_quotients():
    """bin_atom forms the image range from a single-precision reciprocal of the box length and falls back to the double-precision
    quotient within 1e-5 of a whole number: atoms placed so that (position + reach) / L is a whole number +- 1e-7 ... 1e-3 must bin
    exactly like the reference composition says (the features of the periodic call == the oracle's minimum image)."""
    rng = np.random.default_rng(8)
    L = np.array([24.0, 26.0, 25.0], np.float32)
    nv = np.array([30, 30, 30])                          # a grid wider than the box: several images per atom
    origin = np.array([[0.0, 0.0, 0.0]])
    Rp = 5.001
    pos = []
    for k in range(-1, 2):
        for eps in (0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3):
            for ax in range(3):
                p = rng.uniform(2.0, 20.0, 3)
                p[ax] = -Rp - (k + eps) * float(L[ax])                       # (-Rp - p) / L = k + eps
                pos.append(p.copy())
                p[ax] = (nv[ax] - 1) + Rp - (k + eps) * float(L[ax])         # (n - 1 + Rp - p) / L = k + eps
                pos.append(p.copy())
    coords = np.array(pos, np.float32)
    sig = np.zeros((len(coords), 8)); sig[:, 7] = 1.7; sig[::2, 0] = 1.55
    got, err = E.voxelize_lattice(coords, [0, len(coords)], sig, origin, nv, 1.0, box=L[None], prepass_mode=0)
    assert err == 0
    exp = oracle_lattice(coords, np.array([0, len(coords)]), sig, origin, nv, 1.0, L[None])
    assert np.abs(got.astype(np.float64) - exp).max() <= TOL


def test_topology_fixup_jobs_are_per_wide_atom_and_decide_the_shell_exactly():
    """Round 6: a topology call's exact cut-off fix-up runs one job per (item, WIDE atom of the molecule) from
    the handle's list (round 5: one wave per item walked every atom).  Atoms on lattice points nudged by float32 ulps put 30 voxels
    each within ~1e-6 A of the 5 A shell; sigma 1.81 A (channel 1) is re-decided in double: after the topology call NO value of
    that channel is on the other side of the cutoff than the reference's -- which only holds if the fix-up found every wide
    atom of every frame -- and the topology call is bit for bit the plain call."""
    rng = np.random.default_rng(4)
    base = np.stack(np.meshgrid(*[6.0 + 12.0 * np.arange(2)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    frames = []
    for _ in range(3):
        c = base.copy()
        for k in range(len(c)):
            for ax in range(3):
                steps = int(rng.integers(-3, 4))
                for _ in range(abs(steps)):
                    c[k, ax] = np.nextafter(c[k, ax], np.float32(np.inf if steps > 0 else -np.inf))
        frames.append(c)
    n, F = len(base), len(frames)
    coords = np.concatenate(frames)
    sig = np.zeros((n, 8))
    sig[:, 0] = 1.80
    sig[1::2, 1] = 1.81                       # wide atoms: every second atom of the molecule
    sig[::3, 7] = 2.27
    offs = np.arange(F + 1) * n
    origins = np.zeros((F, 3))
    nv = [24, 24, 24]
    topo, err, wide = E.voxelize_lattice_topo(coords, sig, F, origins, nv, 1.0)
    plain, e0 = E.voxelize_lattice(coords, offs, np.tile(sig, (F, 1)), origins, nv, 1.0, prepass_mode=0)
    assert err == 0 and e0 == 0 and wide and np.array_equal(topo, plain)
    exp = oracle_lattice(coords, offs, np.tile(sig, (F, 1)), origins, np.array(nv), 1.0)
    assert np.abs(topo - exp).max() <= TOL
    for c in (1, 7):
        assert not ((topo[:, :, c] == 0) != (exp[:, :, c] == 0)).any(), c
    # the hits recomputed inside k_tail (one wave per value over all atoms, rounds 3-5) instead of by k_exact_redo: the same bits
    inplace, err, _ = E.voxelize_lattice_topo(coords, sig, F, origins, nv, 1.0, exact_redo=-1)
    assert err == 0 and np.array_equal(inplace, topo)
    # a list of FIVE hits (the call has hundreds): the overflow word sends every shell through the in-place pass behind k_exact_redo
    tiny, err, _ = E.voxelize_lattice_topo(coords, sig, F, origins, nv, 1.0, exact_redo=5)
    assert err == 0 and np.array_equal(tiny, topo)
    # a molecule of more than one slice of k_exact_redo (2 048 atoms): the lattice atoms between two halves of filler, wide sigmas on
    # both sides of the slice boundary, so that a re-decided value is the maximum of what two waves found
    fill = rng.uniform(0.5, 23.5, size=(2200, 3)).astype(np.float32)
    big = np.concatenate([fill[:1100], frames[0], fill[1100:]])
    sigb = np.zeros((len(big), 8))
    sigb[:, 0] = 1.5
    sigb[::5, 7] = 2.0                        # wide, all over the molecule
    sigb[1100:1100 + n] = sig
    ob = np.array([0, len(big)])
    t0, err0, w0 = E.voxelize_lattice_topo(big, sigb, 1, origins[:1], nv, 1.0)
    t1, err1, _ = E.voxelize_lattice_topo(big, sigb, 1, origins[:1], nv, 1.0, exact_redo=-1)
    assert err0 == 0 and err1 == 0 and w0 and np.array_equal(t0, t1)
    expb = oracle_lattice(big, ob, sigb, origins[:1], np.array(nv), 1.0)
    assert np.abs(t0 - expb).max() <= TOL
    assert not ((t0[:, :, 7] == 0) != (expb[:, :, 7] == 0)).any()
    # ragged offsets with wide atoms: flagged on the device, and the fix-up jobs do not index the handle with them
    bad = np.array([0, n - 2, 2 * n + 1, 3 * n])
    _, err, _ = E.voxelize_lattice_topo(coords, sig, F, origins, nv, 1.0, atom_offsets=bad)
    assert err & 8
