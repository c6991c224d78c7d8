"""CPU tier, distance_utils row: the float32 oracle (oracle/distance_oracle.c) and the
emulated HIP kernels (csrc/dist_kernels.h through tests/emu) against outputs of the REAL reference's compiled
distance_utils (tests/golden/distance_cases.npz) -- all comparisons are bit-exact."""
import numpy as np
import pytest

from oracle import oracle
from tests import emu_build as E
from tests.cases import golden


@pytest.fixture(scope="module")
def g():
    return golden("distance_cases.npz")


def _groups(g):
    return [list(x) for x in g["groups1"]], [list(x) for x in g["groups2"]]


@pytest.mark.parametrize("impl", ["oracle", "emu"])
def test_dist_trajectory_bit_exact(g, impl):
    f = oracle.dist_trajectory if impl == "oracle" else E.dist_trajectory
    c, b, ch = g["coords"], g["box"], g["chains"]
    for pbc in (0, 1):
        assert np.array_equal(f(c, b, g["sel1"], g["sel2"], ch, False, pbc), g[f"dist_cross_pbc{pbc}"])
        assert np.array_equal(f(c, b, g["sel2"], g["sel2"], ch, True, pbc), g[f"dist_self_pbc{pbc}"])
    assert np.array_equal(f(c, np.zeros_like(b), g["sel1"], g["sel2"], np.zeros(len(ch), np.uint32), False, False),
                          g["dist_cross_zero_box"])


@pytest.mark.parametrize("impl", ["oracle", "emu"])
def test_reductions_bit_exact(g, impl):
    f = oracle.dist_trajectory_reduction if impl == "oracle" else E.dist_reduction
    c, b, m = g["coords"], g["box"], g["masses"]
    g1, g2 = _groups(g)
    for r1 in (0, 1):
        for r2 in (0, 1):
            for pbc in (0, 1):
                got = f(c, b, g1, g2, g["gchains1"], g["gchains2"], False, pbc, m, r1, r2)
                assert np.array_equal(got, g[f"red_{r1}{r2}_pbc{pbc}"]), (r1, r2, pbc)
    assert np.array_equal(f(c, b, g2, g2, g["gchains2"], g["gchains2"], True, True, m, 0, 0), g["red_self"])
    assert np.array_equal(f(c, b, g1, g2[:5], g["gchains1"], g["gchains2"][:5], False, True, m, 0, 1, pairs=True),
                          g["red_pairs"])


@pytest.mark.parametrize("impl", ["oracle", "emu"])
def test_cdist_pdist_bit_exact(g, impl):
    cd, pd = (oracle.cdist, oracle.pdist) if impl == "oracle" else (E.cdist, E.pdist)
    for D in (1, 2, 3, 5):
        assert np.array_equal(cd(g[f"cdist_a{D}"], g[f"cdist_b{D}"]), g[f"cdist_r{D}"])
        assert np.array_equal(pd(g[f"cdist_b{D}"]), g[f"pdist_r{D}"])


def test_contacts_from_squared_distances_match_reference_lists(g):
    """contacts_trajectory (distance_utils.pyx:59-93) == threshold on the squared distances, in loop order."""
    c, b, ch = g["coords"], g["box"], g["chains"]
    d2 = E.dist_trajectory(c, b, g["sel1"], g["sel2"], ch, False, True, squared=True)
    thr = np.float32(12.0) * np.float32(12.0)
    flat, counts = [], []
    n2 = len(g["sel2"])
    for f in range(d2.shape[0]):
        hit = np.nonzero(d2[f] <= thr)[0]
        i, j = np.divmod(hit, n2)
        counts.append(len(hit))
        flat.extend(np.stack([g["sel1"][i], g["sel2"][j]], 1).ravel().tolist())
    assert np.array_equal(counts, g["contacts_counts"]) and np.array_equal(flat, g["contacts_flat"])


def test_device_side_contact_lists_match_reference_lists(g):
    """contacts_trajectory through the count / scan / fill kernels (emulated): the reference's lists, in its order,
    also when a tiny counter budget cuts the frames into several chunks, and for self pairs."""
    c, b, ch = g["coords"], g["box"], g["chains"]
    for budget in (256 << 20, 1):
        for device_sink in (False, True):       # (round 6: the "_dev" form's list stays in one buffer that grows chunk by chunk)
            res = E.contacts_trajectory(c, b, g["sel1"], g["sel2"], ch, False, True, 12.0, budget_bytes=budget, device_sink=device_sink)
            assert np.array_equal([len(x) // 2 for x in res], g["contacts_counts"])
            assert np.array_equal(np.concatenate([np.asarray(x, np.int64) for x in res]), g["contacts_flat"])
    res = E.contacts_trajectory(c, b, g["sel2"], g["sel2"], ch, True, False, 15.0)
    assert np.array_equal([len(x) // 2 for x in res], g["contacts_self_counts"])
    assert np.array_equal(np.concatenate([np.asarray(x, np.int64) for x in res]), g["contacts_self_flat"])
    # more frames than one 64-frame slab, pairs not a multiple of the tile edge, a threshold nothing meets
    rng = np.random.default_rng(9)
    N, F = 40, 150
    xyz = rng.uniform(-12, 12, size=(N, 3, F)).astype(np.float32)
    box = np.full((3, F), 27.0, np.float32)
    chains = (np.arange(N) % 3).astype(np.uint32)
    s1, s2 = np.arange(0, 13, dtype=np.uint32), np.arange(9, 40, dtype=np.uint32)
    for thr in (6.0, 0.01):
        d2 = oracle.dist_trajectory(xyz, box, s1, s2, chains, False, True, squared=True)
        res = E.contacts_trajectory(xyz, box, s1, s2, chains, False, True, thr, budget_bytes=4 * 7 * 64)
        for f in range(F):
            hit = np.nonzero(d2[f] <= np.float32(thr) * np.float32(thr))[0]
            i, j = np.divmod(hit, len(s2))
            assert res[f] == np.stack([s1[i], s2[j]], 1).astype(np.int64).ravel().tolist()


def test_squareform_and_known_answers():
    """The reference's own known-answer tests (tests/test_distance.py:1-28) on the oracle + host squareform."""
    from moleculekit_amd.distance_utils import squareform
    g = golden("distance_cases.npz")
    assert np.array_equal(squareform(g["pdist_r3"]), g["squareform"])
    x = np.array([0, 1, 2], np.float32)[:, None]; y = np.array([3, 4, 5], np.float32)[:, None]
    assert np.allclose(oracle.cdist(x, y), [[3, 4, 5], [2, 3, 4], [1, 2, 3]])
    assert np.allclose(oracle.pdist(np.array([[4, 5], [6, 7], [8, 9]], np.float32)), [2.828427, 5.656854, 2.828427])


def test_emu_larger_random_case_matches_oracle():
    """Tile edges (frames / pairs not multiples of 64), self pairs with n1 != n2, NaN from a zero box."""
    rng = np.random.default_rng(3)
    N, F = 150, 70
    c = rng.uniform(-30, 30, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(15, 25, size=(3, F)).astype(np.float32)
    b[:, 5] = 0.0                                           # pbc with a zero box -> NaN, like the reference
    ch = rng.integers(0, 3, size=N).astype(np.uint32)
    s1 = np.arange(0, 90, dtype=np.uint32); s2 = np.arange(40, 150, dtype=np.uint32)
    for selfd, a, bb in ((False, s1, s2), (True, s2, s2), (True, s1[:7], s2[:20])):
        exp = oracle.dist_trajectory(c, b, a, bb, ch, selfd, True)
        for avoid in (0, 1):                                # the block-per-frame kernel (round 5), then the tile kernels it stands in front of
            got = E.dist_trajectory(c, b, a, bb, ch, selfd, True, avoid=avoid)
            assert np.array_equal(got, exp, equal_nan=True), (selfd, avoid)
        assert np.isnan(exp[5]).any()


def _half_box_case():
    """Separations sitting on, and one ulp either side of, half the box edge (the rounding boundary of the image
    shift `round(d / box)`, distance_utils.pyx:49-52), for boxes whose reciprocal is inexact: only the correctly
    rounded quotient rounds all of them like the reference."""
    boxes = np.array([10.0, 33.3, 66.9, 7.123456, 100.0, 1e-3, 3.0e4], np.float32)
    F = len(boxes)
    seps = []
    for k in (0.5, 1.5, 2.5):
        h = (boxes * np.float32(k)).astype(np.float32)
        seps += [h, np.nextafter(h, np.float32(0)), np.nextafter(h, np.float32(np.inf)),
                 np.nextafter(np.nextafter(h, np.float32(0)), np.float32(0))]
    n = 1 + 2 * len(seps)
    c = np.zeros((n, 3, F), np.float32)
    for i, s in enumerate(seps):
        c[1 + 2 * i, 0] = s            # +x
        c[2 + 2 * i, 1] = -s           # -y
        c[2 + 2 * i, 2] = s * np.float32(0.75)
    b = np.tile(boxes[None, :], (3, 1)).astype(np.float32)
    ch = np.arange(n, dtype=np.uint32)          # every atom its own chain: always wrapped
    return c, b, ch, np.array([0], np.uint32), np.arange(1, n, dtype=np.uint32)


def test_image_shift_at_the_half_box_boundary_is_bit_exact():
    c, b, ch, s1, s2 = _half_box_case()
    got = E.dist_trajectory(c, b, s1, s2, ch, False, True)
    assert np.array_equal(got, oracle.dist_trajectory(c, b, s1, s2, ch, False, True))


def test_rectangular_kernel_shapes_and_modes_match_oracle():
    """Round 4: dist_trajectory without selfdist takes k_dist_rect (a block keeps its 64 second atoms in registers and walks
    8 first atoms; no pair table).  Shapes around its tile edges -- n2 below / at / above 64 and 128, n1 below / at / above the
    8 first atoms of a block, frames not a multiple of 64 -- with and without pbc, squared and not, unsorted and repeated
    atom indices: the kernel source on the host emulator against the oracle, bit for bit."""
    rng = np.random.default_rng(11)
    N, F = 260, 67
    c = rng.uniform(-30, 30, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(15, 25, size=(3, F)).astype(np.float32)
    ch = rng.integers(0, 4, size=N).astype(np.uint32)
    for n1, n2 in ((1, 1), (7, 63), (8, 64), (9, 65), (17, 130), (3, 200)):
        s1 = rng.integers(0, N, size=n1).astype(np.uint32)
        s2 = rng.integers(0, N, size=n2).astype(np.uint32)
        for pbc in (False, True):
            for sq in (False, True):
                got = E.dist_trajectory(c, b, s1, s2, ch, False, pbc, squared=sq, avoid=3)       # neither the frame nor the row kernel: k_dist_rect
                exp = oracle.dist_trajectory(c, b, s1, s2, ch, False, pbc, squared=sq)
                assert got.shape == (F, n1 * n2) and np.array_equal(got, exp), (n1, n2, pbc, sq)


def test_block_per_frame_kernel_shapes_and_modes_match_oracle():
    """Round 5: k_dist_frame -- a block per frame (or slice of a frame's pairs) stages both selections in LDS and walks the
    rectangular pair list in memory order, four consecutive pairs per lane: four consecutive frames per block where both selections fit 32 KB of LDS (one 16-byte load per atom and axis when the
    frame count is a multiple of four), rows shorter and longer than a lane's four pairs and
    than a block's step of 1 024, lists that end inside a lane's four pairs, frame rows that do and do not start on 16 bytes, one
    frame and several slices per frame, more than 1 024 atoms (the big LDS tier), pbc on and off, squared and not, a zero box
    edge and NaN coordinates -- the kernel source on the host emulator against the oracle, bit for bit.  (Triangular lists ride
    along: they keep the pair-table kernel whatever is avoided.)"""
    rng = np.random.default_rng(41)
    N = 1500
    ch = rng.integers(0, 4, size=N).astype(np.uint32)
    for F in (1, 5, 8):                                       # (8: four frames per block through 16-byte loads; 5: a ragged last group)
        c = rng.uniform(-30, 30, size=(N, 3, F)).astype(np.float32)
        b = rng.uniform(15, 25, size=(3, F)).astype(np.float32)
        if F > 1:
            b[1, 3] = 0.0
            c[3, 0, 1] = np.nan; c[7, 1, 2] = 3e38
        cases = [(False, 1, 1), (False, 7, 3), (False, 30, 30), (False, 300, 30), (False, 5, 1030), (False, 3, 4100 // 3), (False, 41, 101),
                 (False, 1030, 2), (True, 5, 5), (True, 61, 61), (True, 12, 90)]
        for selfd, n1, n2 in cases:
            s2 = rng.integers(0, N, size=n2).astype(np.uint32)
            s1 = s2[:n1].copy() if (selfd and n1 <= n2) else rng.integers(0, N, size=n1).astype(np.uint32)
            for pbc in (False, True):
                for sq in ((False, True) if n1 * n2 < 2000 else (False,)):
                    exp = oracle.dist_trajectory(c, b, s1, s2, ch, selfd, pbc, squared=sq)
                    got = E.dist_trajectory(c, b, s1, s2, ch, selfd, pbc, squared=sq, avoid=2)     # (never the row kernel: the frame kernel)
                    assert got.shape == exp.shape and np.array_equal(got, exp, equal_nan=True), (F, selfd, n1, n2, pbc, sq)
    # more atoms than the small LDS tier holds; a list that the row kernel would take, through the frame kernel
    c = rng.uniform(-30, 30, size=(N, 3, 2)).astype(np.float32)
    b = rng.uniform(15, 25, size=(3, 2)).astype(np.float32)
    s1, s2 = np.arange(0, 40, dtype=np.uint32), np.arange(100, 1400, dtype=np.uint32)     # 1 340 atoms: one frame per block
    assert np.array_equal(E.dist_trajectory(c, b, s1, s2, ch, False, True, avoid=2), oracle.dist_trajectory(c, b, s1, s2, ch, False, True))


def test_row_kernel_shapes_and_modes_match_oracle():
    """Round 4: rows of >= 64 second atoms are written by a wave per frame from selections turned frame-major first
    (k_sel_to_frames + k_dist_rows).  Which (n1, n2) take it and with how many second atoms per lane is decided in
    run_dist_trajectory (>= 80 % of the lanes busy, the turned selections at most a quarter of the result); shapes on both
    sides of every decision and around the kernels' edges -- n2 at / around 64, 128, 256 and beyond, first atoms below / at /
    above a wave's chunk of 16, frames not a multiple of 64 (the turning kernel's tiles), one frame -- with and without pbc,
    squared and not, unsorted and repeated atom indices; everything against the oracle, bit for bit."""
    import ctypes
    jpl = lambda n1, n2, F: E.lib().emu_dist_rows_jpl(ctypes.c_longlong(n1), ctypes.c_longlong(n2), ctypes.c_longlong(F))
    # (the last three: four second atoms per lane with rows that are NOT a multiple of four long -- 16-byte stores at 4-byte
    #  alignment and a lane whose four straddle the end of the row, round 5)
    shapes = ((40, 52), (40, 64), (15, 70), (24, 128), (17, 129), (33, 200), (20, 256), (30, 500), (50, 260), (16, 128),
              (20, 254), (20, 511), (25, 253))
    assert [jpl(*s, 67) for s in shapes] == [1, 1, 0, 2, 0, 0, 4, 4, 1, 0, 4, 4, 4]
    assert jpl(1, 64, 67) == 0 and jpl(200, 500, 2048) == 4 and jpl(300, 30, 100) == 0       # (too few pairs; the bench leg; short rows)
    rng = np.random.default_rng(12)
    N = 300
    ch = rng.integers(0, 4, size=N).astype(np.uint32)
    for F in (1, 67):
        c = rng.uniform(-30, 30, size=(N, 3, F)).astype(np.float32)
        b = rng.uniform(15, 25, size=(3, F)).astype(np.float32)
        for n1, n2 in shapes:
            s1 = rng.integers(0, N, size=n1).astype(np.uint32)
            s2 = rng.integers(0, N, size=n2).astype(np.uint32)
            for pbc in (False, True):
                for sq in (False, True):
                    exp = oracle.dist_trajectory(c, b, s1, s2, ch, False, pbc, squared=sq)
                    for avoid in (16, 0):                        # the row kernel wherever it applies (rounds 4-5), then round 6's choice among the three
                        got = E.dist_trajectory(c, b, s1, s2, ch, False, pbc, squared=sq, avoid=avoid)
                        assert got.shape == (F, n1 * n2) and np.array_equal(got, exp), (F, n1, n2, pbc, sq, avoid)
    # a zero box edge (the reference divides by it: NaN), NaN and huge coordinates: the extraordinary roots' branch
    F = 5
    c = rng.uniform(-30, 30, size=(N, 3, F)).astype(np.float32)
    c[3, 0, 1] = np.nan; c[7, 1, 2] = 3e38; c[9, 2, 0] = -3e38; c[11] = c[12]
    b = rng.uniform(15, 25, size=(3, F)).astype(np.float32); b[1, 3] = 0.0
    s1 = np.arange(0, 40, dtype=np.uint32); s2 = np.arange(0, 128, dtype=np.uint32)
    assert jpl(40, 128, F) == 2
    for pbc in (False, True):
        got = E.dist_trajectory(c, b, s1, s2, ch, False, pbc)
        exp = oracle.dist_trajectory(c, b, s1, s2, ch, False, pbc)
        assert np.array_equal(got, exp, equal_nan=True) and np.array_equal(np.isnan(got), np.isnan(exp))


def test_row_kernel_image_shift_is_bit_exact_at_every_boundary():
    """The image shift d - b * round(d / b) (distance_utils.pyx:49-51) through the ROW kernel (rows of 64 / 128 second atoms):
    coordinates wrapped into the box (every separation within one box length: the usual trajectory); separations ON half
    the box edge and one and two ulps either side, on the box edge itself and either side of it, for boxes whose half or
    reciprocal is inexact in binary; frames with a zero, an infinite, a NaN, a negative, a denormal box edge; signed zeros
    among the separations; one second atom in the first atoms' chain (never shifted).  Against the oracle, bit for bit.
    (Written for a shift without the quotient -- d -+ b iff |d| >= b / 2 where |d| <= b -- that was correct on all of this
    and 3-5 % slower on the GPU: docs/EXPERIMENTS_r4.md.  The cases stay.)"""
    import ctypes
    rng = np.random.default_rng(31)
    jpl = lambda n1, n2, F: E.lib().emu_dist_rows_jpl(ctypes.c_longlong(n1), ctypes.c_longlong(n2), ctypes.c_longlong(F))
    boxes = np.array([10.0, 33.3, 66.9, 7.123456, 100.0, 3.0e4, 1.1754944e-38 * 3, 1e-3], np.float32)
    F = len(boxes)
    b = np.tile(boxes[None, :], (3, 1)).astype(np.float32)
    # (1) wrapped coordinates, random chains
    for n1, n2 in ((40, 64), (24, 128)):
        assert jpl(n1, n2, F) in (1, 2)
        N = n1 + n2
        c = (rng.random((N, 3, F)).astype(np.float32) * boxes[None, None, :]).astype(np.float32)
        ch = rng.integers(0, 3, size=N).astype(np.uint32)
        s1, s2 = np.arange(n1, dtype=np.uint32), np.arange(n1, N, dtype=np.uint32)
        for avoid in (16, 0):                                  # the row kernel (rounds 4-5: wherever it applies), then round 6's choice (the tile kernel)
            got = E.dist_trajectory(c, b, s1, s2, ch, False, True, avoid=avoid)
            assert np.array_equal(got, oracle.dist_trajectory(c, b, s1, s2, ch, False, True)), avoid
    # (2) separations at the boundaries: first atoms at the origin, second atoms at +-k * box (k = 0.5, 1.0) and neighbours
    n1, n2 = 40, 64
    ch = np.concatenate([np.zeros(n1, np.uint32), np.ones(n2, np.uint32)])     # every pair across chains: always shifted
    ch[n1 + 3] = 0                                                             # (but one second atom: never)
    s1, s2 = np.arange(n1, dtype=np.uint32), np.arange(n1, n1 + n2, dtype=np.uint32)
    for beyond in (False, True):       # False: every separation <= the box edge; True: some just beyond it
        seps = []
        for k in (0.5, 1.0):
            h = (boxes * np.float32(k)).astype(np.float32)
            lo1, hi1 = np.nextafter(h, np.float32(0)), np.nextafter(h, np.float32(np.inf))
            seps += [h, lo1, np.nextafter(lo1, np.float32(0))]
            if k == 0.5 or beyond:
                seps += [hi1, np.nextafter(hi1, np.float32(np.inf))]
        seps += [np.zeros(F, np.float32), (boxes * np.float32(0.25)).astype(np.float32), (boxes * np.float32(0.75)).astype(np.float32)]
        c = np.zeros((n1 + n2, 3, F), np.float32)
        c[1:n1:7] = np.float32(-0.0)                                           # first atoms at +0, a few at -0
        for j in range(n2):
            s = seps[j % len(seps)] * np.float32(-1.0 if (j // len(seps)) % 2 else 1.0)
            c[n1 + j, j % 3] = s
            c[n1 + j, (j + 1) % 3] = seps[(j + 5) % len(seps)] * np.float32(0.5)
        for avoid in (16, 0):
            got = E.dist_trajectory(c, b, s1, s2, ch, False, True, avoid=avoid)
            assert np.array_equal(got, oracle.dist_trajectory(c, b, s1, s2, ch, False, True)), (beyond, avoid)
    # (3) boxes that are not ordinary positive numbers: the general path, NaN where the reference gives NaN
    b2 = b.copy()
    b2[0, 0] = 0.0; b2[1, 1] = np.inf; b2[2, 2] = np.nan; b2[0, 3] = -7.0; b2[1, 4] = 1e-42
    c3 = (rng.random((n1 + n2, 3, F)).astype(np.float32) * np.float32(9.0)).astype(np.float32)
    exp = oracle.dist_trajectory(c3, b2, s1, s2, ch, False, True)
    for avoid in (16, 0):
        got = E.dist_trajectory(c3, b2, s1, s2, ch, False, True, avoid=avoid)
        assert np.array_equal(got, exp, equal_nan=True) and np.array_equal(np.isnan(got), np.isnan(exp)), avoid


# ------------------------------------------------------------------------------------------------
# the reference's own MetricDistance projections on its real trajectory (tests/metricdistance_real.py)
# ------------------------------------------------------------------------------------------------
class _OracleFns:
    @staticmethod
    def dist_trajectory(coords, box, s1, s2, chains, selfdist, pbc, res):
        res[...] = oracle.dist_trajectory(coords, box, s1, s2, chains, selfdist, pbc)

    @staticmethod
    def dist_trajectory_reduction(coords, box, g1, g2, c1, c2, selfdist, pbc, masses, r1, r2, res):
        res[...] = oracle.dist_trajectory_reduction(coords, box, g1, g2, c1, c2, selfdist, pbc, masses, r1, r2)


    @staticmethod
    def dist_trajectory_reduction_pairs(coords, box, g1, g2, c1, c2, pbc, masses, r1, r2, res):
        res[...] = oracle.dist_trajectory_reduction(coords, box, g1, g2, c1, c2, False, pbc, masses, r1, r2, pairs=True)


def test_oracle_on_the_known_answers_of_the_references_metricdistance_tests():
    """tests/metricdistance_known.py: the analytic molecules, the four 3PTB numbers (8.978174, 3.8286476, 2.8153415, the distance of the
    centres of mass), pairs mode, the three meanings of `periodic` on the trajectory and `truncate` -- every call the reference's
    projections made, replayed through the oracle: bit for bit the compiled reference, and the reference tests' own assertions."""
    from tests import metricdistance_known as K, metricdistance_real as M
    g = K.load()
    traj = M.read_trajectory(M.load())
    with_answer = sum(K.verify(K.replay(_OracleFns, g, key, traj), g, key) for key in map(str, g["names"]))
    assert len(g["names"]) == 22 and with_answer == 17


def test_oracle_on_the_reference_held_metricdistance_projections():
    """Host XTC reader -> oracle == the compiled reference bit for bit, and within the reference's own 1e-3 of the arrays the
    reference holds (they were written by an older build: 7.6e-6 / 3.8e-6 away from today's reference too)."""
    from tests import metricdistance_real as M
    g = M.load()
    coords, box = M.read_trajectory(g)
    for key in M.KEYS:
        exact, worst = M.check(M.run_projection(_OracleFns, coords, box, g, key), g, key)
        assert exact, key
        assert worst is None or worst < 1e-5, (key, worst)
    # the held contact matrix (contacts.npy beside distances.npy; no test of the reference reads it any more, and it is in the
    # pair order of an older build: ligand atom-major): the `metric="contacts"` post-processing of the drivers, threshold 8
    d = M.run_projection(_OracleFns, coords, box, g, "distances")
    n1, n2 = len(g["distances_sel1"]), len(g["distances_sel2"])
    assert np.array_equal((d <= 8).reshape(-1, n1, n2).transpose(0, 2, 1).reshape(d.shape[0], -1), g["contacts_held"])


# ------------------------------------------------------------------------------------------------
# round 6: k_dist_reduction_closest (first-group atoms in registers, packed arithmetic, accumulated exactness test)
# ------------------------------------------------------------------------------------------------
def _ragged_groups(rng, n_atoms, ng, lo, hi):
    return [rng.choice(n_atoms, int(rng.integers(lo, hi + 1)), replace=False).tolist() for _ in range(ng)]


@pytest.mark.parametrize("selfdist,pairs", [(False, False), (True, False), (False, True)])
def test_closest_reduction_kernel_every_block_size_is_the_oracle(selfdist, pairs):
    """Ragged groups of 1 ... 19 atoms (padding slots, several passes per first group), 70 frames (a frame tile and a ragged
    one), enough group pairs for several tiles and for waves whose stretch crosses first groups: 4 and 8 first atoms in
    registers, the generic kernel and the oracle agree bit for bit, periodic with mixed chains and open."""
    rng = np.random.default_rng(601 + 2 * selfdist + pairs)
    N, F = 150, 70
    coords = rng.uniform(-30, 30, size=(N, 3, F)).astype(np.float32)
    box = rng.uniform(22, 31, size=(3, F)).astype(np.float32)
    masses = np.ones(N, np.float32)
    g1 = _ragged_groups(rng, N, 13, 1, 19)
    g2 = g1 if selfdist else _ragged_groups(rng, N, 13 if pairs else 11, 1, 12)
    ch1 = rng.integers(0, 3, len(g1)).astype(np.uint32)
    ch2 = ch1 if selfdist else rng.integers(0, 3, len(g2)).astype(np.uint32)
    for pbc in (True, False):
        want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, selfdist, pbc, masses, 0, 0, pairs=pairs)
        for block in (4, 8, -1, 0, 104, 108, -2):   # (+ 100: blocks of four waves; -2: the few-frame kernel, lanes along second groups)
            got = E.dist_reduction(coords, box, g1, g2, ch1, ch2, selfdist, pbc, masses, 0, 0, pairs=pairs, block=block)
            assert np.array_equal(got, want), (pbc, block)
    # 64-bit row addressing (what a trajectory of more than 4 GiB takes): same bits
    got = E.dist_reduction(coords, box, g1, g2, ch1, ch2, selfdist, True, masses, 0, 0, pairs=pairs, block=4, n_atoms=1 << 40)
    assert np.array_equal(got, oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, selfdist, True, masses, 0, 0, pairs=pairs))


def _image_integer_traps(rng, n):
    """(box length b, separation d) float32 pairs for which rndne(fl(d * fl(1/b))) -- the kernels' fast image integer -- is NOT
    the reference's round(fl(d / b)) AND the shifted separations differ: quotients on or within an ulp of a half-integer."""
    f32, out = np.float32, []
    while len(out) < n:
        b = f32(rng.uniform(15, 40)); k = int(rng.integers(-3, 3))
        d = f32((k + 0.5) * float(b))
        for _ in range(int(rng.integers(0, 3))):
            d = np.nextafter(d, f32(np.inf if rng.integers(2) else -np.inf), dtype=np.float32)
        q_fast = f32(d * (f32(1) / b)); r_fast = np.rint(q_fast)
        q_ref = float(f32(d / b)); r_ref = np.sign(q_ref) * np.floor(abs(q_ref) + 0.5)             # C round(): half away from zero
        if r_fast != r_ref and abs(f32(d - f32(b * f32(r_fast)))) != abs(f32(d - f32(b * f32(r_ref)))):
            out.append((b, d))
    return out


def test_closest_reduction_kernel_redoes_stretches_near_half_a_box_exactly():
    """Separations for which the fast image integer differs from the reference's round(d / b) (found by search: quotients on
    half-integers, where round-half-even and round-half-away part, or an ulp beside them): the accumulated risk sends those
    stretches through the pair-by-pair path with its correctly rounded divisions; bit for bit the oracle.  Single-atom groups,
    so that the trapped pair IS the minimum (tests/test_distance_cpu.py was mutation-checked: without the redo this fails)."""
    rng = np.random.default_rng(77)
    F = 64
    traps = _image_integer_traps(rng, F)
    N = 24
    coords = rng.uniform(0, 12, size=(N, 3, F)).astype(np.float32)
    box = np.empty((3, F), np.float32)
    for f, (b, d) in enumerate(traps):
        ax = f % 3
        box[:, f] = [np.float32(41.3), np.float32(37.9), np.float32(44.1)]
        box[ax, f] = b
        coords[1, :, f] = coords[0, :, f]                      # atom 1 = atom 0 shifted by the trap along one axis
        coords[1, ax, f] = np.float32(coords[0, ax, f] - d)
    g1 = [[0], [2, 3, 4, 5, 6]]
    g2 = [[1], [7, 8, 9]]
    ch1, ch2 = np.zeros(2, np.uint32), np.ones(2, np.uint32)
    masses = np.ones(N, np.float32)
    want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, True, masses, 0, 0)
    fast = []                                                  # what the fast integer alone would give for the trapped pair
    for f, (b, d) in enumerate(traps):
        ax = f % 3
        dd = np.float32(coords[0, ax, f] - coords[1, ax, f])
        fast.append(dd == d)
    assert sum(fast) > F // 2                                  # (the subtraction reproduces the trap's separation in most frames)
    for block in (4, 8, -2):
        assert np.array_equal(E.dist_reduction(coords, box, g1, g2, ch1, ch2, False, True, masses, 0, 0, block=block), want)


def test_closest_reduction_kernel_nan_and_infinite_semantics():
    """The reference's `if dist2 < mindist or mindist < 0` keeps a NaN only when the FIRST atom pair produced it
    (distance_utils.pyx:268-274); an infinite coordinate gives inf or NaN like the reference; a zero box with pbc too."""
    rng = np.random.default_rng(5)
    N, F = 30, 64
    coords = rng.uniform(0, 15, size=(N, 3, F)).astype(np.float32)
    box = np.full((3, F), 25.0, np.float32)
    coords[4, 1, ::3] = np.nan            # first atom of group 0 of side 1: every result of that row is NaN in those frames
    coords[9, 2, ::5] = np.nan            # a LATER atom of a group: ignored by the minimum
    coords[12, 0, ::7] = np.inf
    box[:, 10] = 0.0                      # pbc with a zero box: the reference divides by zero
    g1 = [[4, 5, 6], [7, 8, 9, 10, 11], [12, 13]]
    g2 = [[14, 15, 16, 17], [18, 9], [20, 21, 22, 23, 24, 25, 26, 27, 28]]
    ch1, ch2 = np.zeros(3, np.uint32), np.ones(3, np.uint32)
    masses = np.ones(N, np.float32)
    with np.errstate(all="ignore"):
        for pbc in (True, False):
            want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, pbc, masses, 0, 0)
            assert np.isnan(want).any() and np.isfinite(want).any()
            for block in (4, 8, -1, -2):
                got = E.dist_reduction(coords, box, g1, g2, ch1, ch2, False, pbc, masses, 0, 0, block=block)
                assert np.array_equal(got, want, equal_nan=True), (pbc, block)
        # mixed chains: in the few-frame kernel (-2) the lanes whose pair does NOT wrap ride along the wrapping walk with 1 / box = 0 --
        # an infinite separation there must stay infinite (0 * inf is NaN), a first-pair NaN must stay the reference's
        coords[20, 2, 2::7] = -np.inf
        g1m, g2m = [[12], [4, 5, 6], [13, 12]], [[14, 15], [18, 9], [20], [21, 22]]
        ch1m, ch2m = np.zeros(3, np.uint32), np.array([0, 1, 0, 0], np.uint32)
        want = oracle.dist_trajectory_reduction(coords, box, g1m, g2m, ch1m, ch2m, False, True, masses, 0, 0)
        assert np.isinf(want).any() and np.isnan(want).any() and np.isfinite(want).any()
        for block in (-2, 8, -1):
            got = E.dist_reduction(coords, box, g1m, g2m, ch1m, ch2m, False, True, masses, 0, 0, block=block)
            assert np.array_equal(got, want, equal_nan=True), block


def test_reductions_of_few_frames_take_lanes_along_the_second_groups():
    """k_dist_reduction_few (round 6): calls of up to 8 frames (16 when periodic) -- one structure's residue-contact map -- run their
    lanes along the second groups.  The library's own choice (block 0) at 1, 3 and 8 frames: more second groups than one block holds (several
    blocks along a row, selfdist rows that skip the blocks in front of their diagonal), a first group larger than one LDS pass
    (300 atoms > 256), single-atom groups, every reduction mode (closest / centre of mass on either side), periodic with mixed
    chains and open; bit for bit the oracle, and the kernels whose lanes are frames (block 8 / -1) on the same call."""
    rng = np.random.default_rng(811)
    N = 700
    masses = rng.uniform(1, 32, size=N).astype(np.float32)
    big = rng.choice(N, 300, replace=False).tolist()
    for F in (1, 3, 8):
        coords = rng.uniform(-20, 20, size=(N, 3, F)).astype(np.float32)
        box = rng.uniform(22, 31, size=(3, F)).astype(np.float32)
        g1 = [big] + _ragged_groups(rng, N, 4, 1, 9)
        g2 = _ragged_groups(rng, N, 290, 1, 4)
        ch1 = rng.integers(0, 3, len(g1)).astype(np.uint32); ch2 = rng.integers(0, 3, len(g2)).astype(np.uint32)
        for r1, r2 in ((0, 0), (1, 0), (0, 1), (1, 1)) if F == 3 else ((0, 0),):
            for pbc in (True, False):
                want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, pbc, masses, r1, r2)
                assert np.array_equal(E.dist_reduction(coords, box, g1, g2, ch1, ch2, False, pbc, masses, r1, r2), want), (F, r1, r2, pbc)
        gs = _ragged_groups(rng, N, 270, 1, 3)
        chs = rng.integers(0, 4, len(gs)).astype(np.uint32)
        want = oracle.dist_trajectory_reduction(coords, box, gs, gs, chs, chs, True, True, masses, 0, 0)
        assert np.array_equal(E.dist_reduction(coords, box, gs, gs, chs, chs, True, True, masses, 0, 0), want), F
        if F == 1:
            assert np.array_equal(E.dist_reduction(coords, box, gs, gs, chs, chs, True, True, masses, 0, 0, block=8), want)
            assert np.array_equal(E.dist_reduction(coords, box, gs, gs, chs, chs, True, True, masses, 0, 0, block=-1), want)


@pytest.mark.parametrize("D", [2, 3])
def test_cdist_pdist_row_kernels_ragged_edges_bit_exact(D):
    """Round 6: k_cdist_rows / k_pdist_rows (four second points per lane, 16-byte stores at any alignment): several blocks along both
    axes, row lengths that are not multiples of four, blocks that the diagonal of the condensed triangle crosses; the oracle's bits."""
    rng = np.random.default_rng(40 + D)
    a = rng.normal(0, 15, size=(37, D)).astype(np.float32)
    b = rng.normal(0, 15, size=(1031, D)).astype(np.float32)
    assert np.array_equal(E.cdist(a, b), oracle.cdist(a, b))
    assert np.array_equal(E.cdist(b[:5], a[:3]), oracle.cdist(b[:5], a[:3]))
    c = rng.normal(0, 15, size=(1100, D)).astype(np.float32)
    c[7] = c[3]                                                  # a zero distance
    assert np.array_equal(E.pdist(c), oracle.pdist(c))
    for n in (2, 3, 5, 1025):
        assert np.array_equal(E.pdist(c[:n]), oracle.pdist(c[:n])), n


def _row_kernel_trap_case(mixed, seed=9):
    """16 x 250 pairs x 6 frames -- rows long enough for the row kernel -- in which second atoms sit at image-integer traps from first
    atoms (rndne(d * fl(1/b)) != round(d / b)), one frame has a zero box, one coordinate is inf and one NaN; `mixed`: chain ids vary among
    the second atoms (per-pair wrap flags), else the reference's periodic="selections" ids (1 / 2)."""
    rng = np.random.default_rng(seed)
    n1, n2, F = 16, 250, 6                                     # (250 of 256 lanes x 4: the row kernel with four second atoms per lane)
    N = n1 + n2
    c = rng.uniform(0, 14, size=(N, 3, F)).astype(np.float32)
    b = np.empty((3, F), np.float32)
    traps = _image_integer_traps(rng, 15)
    for f in range(F):
        b[:, f] = [np.float32(43.7), np.float32(39.1), np.float32(47.3)]
    for t, (bl, d) in enumerate(traps):                          # one trap per (frame, axis): its box length is that trap's
        f, ax, i, j = (0, 1, 2, 3, 5)[t // 3], t % 3, t % n1, n1 + (17 * t) % n2
        b[ax, f] = bl
        c[i, ax, f] = 0.0                                        # (so that first - second is EXACTLY the trap's separation)
        c[j, :, f] = c[i, :, f]
        c[j, ax, f] = -d
    b[:, 4] = 0.0                                                # pbc with a zero box: NaN / inf like the reference
    c[n1 + 5, 0, 2] = np.inf
    c[n1 + 9, 1, 3] = np.nan
    ch = np.ones(N, np.uint32); ch[n1:] = 2
    if mixed:
        ch = rng.integers(0, 3, N).astype(np.uint32)
    return c, b, ch, np.arange(n1, dtype=np.uint32), np.arange(n1, N, dtype=np.uint32)


@pytest.mark.parametrize("mixed", [False, True])
def test_periodic_row_kernel_image_integers_are_the_references(mixed):
    """k_dist_rows' periodic rows (four second atoms per lane): image integers at traps (the reciprocal-multiply shortcut alone would get
    them wrong: mutation-checked in round 6 against a packed variant of the kernel whose redo path was switched off), one chain id among
    the second atoms and mixed ids, a zero box, inf / NaN coordinates; with and without the 16-byte stores; squared distances too."""
    c, b, ch, s1, s2 = _row_kernel_trap_case(mixed)
    with np.errstate(all="ignore"):
        for sq in (False, True):
            want = oracle.dist_trajectory(c, b, s1, s2, ch, False, True, squared=sq)
            for avoid in (1, 1 | 8, 3):                          # (not the block-per-frame kernel: the row kernel, with / without 16-byte stores;
                #  3: the rectangular tile kernel -- round 6: its pairs are packed behind the accumulated test as well, mutation-checked)
                got = E.dist_trajectory(c, b, s1, s2, ch, False, True, squared=sq, avoid=avoid)
                assert np.array_equal(got, want, equal_nan=True), (sq, avoid)
        assert np.isnan(want).any()                              # (the zero box, the NaN coordinate; an inf coordinate wraps to NaN as well)


def _pair_table_trap_case(mixed, seed=21):
    """40 atoms x 70 frames for the PAIR-TABLE walk (selfdist: k_dist_pairs, and the contact kernels on the same walk): in every frame one
    pair sits at an image-integer trap (rndne(d * fl(1/b)) != round(d / b)), one frame has a zero box, one coordinate is inf and one NaN;
    every atom its own chain (all pairs wrap: the packed batches), or -- `mixed` -- three chains (batches in which some pairs wrap)."""
    rng = np.random.default_rng(seed)
    N, F = 40, 70
    c = rng.uniform(0, 14, size=(N, 3, F)).astype(np.float32)
    b = np.empty((3, F), np.float32)
    traps = _image_integer_traps(rng, F)
    where = []
    for f, (bl, d) in enumerate(traps):
        b[:, f] = [np.float32(43.7), np.float32(39.1), np.float32(47.3)]
        ax, i = f % 3, int(rng.integers(0, N - 1))
        j = int(rng.integers(i + 1, N))
        if mixed:                                                # (a pair of different chains: it wraps)
            while (i % 3) == (j % 3):
                j = j + 1 if j + 1 < N else i + 1
        b[ax, f] = bl
        c[i, ax, f] = 0.0                                        # (so that first - second is EXACTLY the trap's separation)
        c[j, :, f] = c[i, :, f]
        c[j, ax, f] = -d
        where.append((i, j))
    b[:, 11] = 0.0
    c[7, 0, 12] = np.inf
    c[9, 1, 13] = np.nan
    ch = (np.arange(N) % 3 if mixed else np.arange(N)).astype(np.uint32)
    return c, b, ch, np.arange(N, dtype=np.uint32), where


@pytest.mark.parametrize("mixed", [False, True])
def test_pair_table_walk_packed_batches_redo_image_integer_traps(mixed):
    """for_pair_run (round 6): batches of four wrapping pairs run in packed arithmetic behind one accumulated test of their image
    integers; a batch holding a trap must be redone pair by pair (mutation-checked: with the redo switched off this fails in most frames),
    batches with pairs of one chain among them never take the packed form.  selfdist (the pair table), distances and squares, and the
    contact lists of the same walk."""
    c, b, ch, sel, where = _pair_table_trap_case(mixed)
    with np.errstate(all="ignore"):
        for sq in (False, True):
            want = oracle.dist_trajectory(c, b, sel, sel, ch, True, True, squared=sq)
            got = E.dist_trajectory(c, b, sel, sel, ch, True, True, squared=sq)
            assert np.array_equal(got, want, equal_nan=True), sq
        assert np.isnan(want).any()
        # what the fast integer alone would have given differs from the reference at the trapped pair of most frames
        f32, differ = np.float32, 0
        for f, (i, j) in enumerate(where):
            if f in (11, 12, 13):
                continue
            ax = f % 3
            d, bl = f32(c[i, ax, f] - c[j, ax, f]), b[ax, f]
            r_fast = np.rint(f32(d * (f32(1) / bl)))
            q = float(f32(d / bl)); r_ref = np.sign(q) * np.floor(abs(q) + 0.5)
            differ += r_fast != r_ref
        assert differ > 40
        d2 = oracle.dist_trajectory(c, b, sel, sel, ch, True, True, squared=True)
        iu, ju = np.triu_indices(len(sel), 1)
        for avoid in (1, 0, 5 << 8):                             # the pair-table walk; the rectangular kernels' triangular form (round 6), groups of 1 and 5 rows
            res = E.contacts_trajectory(c, b, sel, sel, ch, True, True, 9.0, avoid=avoid)
            for f in range(c.shape[2]):
                hit = np.nonzero(d2[f] <= f32(81.0))[0]
                assert res[f] == np.stack([sel[iu[hit]], sel[ju[hit]]], 1).astype(np.int64).ravel().tolist(), (f, avoid)


def _contact_lists(d2, s1, s2, thr):
    out = []
    for f in range(d2.shape[0]):
        hit = np.nonzero(d2[f] <= np.float32(thr) * np.float32(thr))[0]
        i, j = np.divmod(hit, len(s2))
        out.append(np.stack([s1[i], s2[j]], 1).astype(np.int64).ravel().tolist())
    return out


def _rect_contact_case(ids, n2, n1=21, F=70, seed=31):
    """n1 x n2 atoms x F frames for the rectangular contact kernel: an image-integer trap in every frame (rndne(d * fl(1/b)) !=
    round(d / b)), a zero box, inf / NaN coordinates; chain ids that make every pair wrap ("selections": 1 / 2), some ("chains"), none ("one")."""
    rng = np.random.default_rng(seed)
    N = n1 + n2
    c = rng.uniform(0, 14, size=(N, 3, F)).astype(np.float32)
    b = np.empty((3, F), np.float32)
    traps = _image_integer_traps(rng, F)
    for f, (bl, d) in enumerate(traps):
        b[:, f] = [np.float32(43.7), np.float32(39.1), np.float32(47.3)]
        ax, i, j = f % 3, f % n1, n1 + (17 * f) % n2
        b[ax, f] = bl
        c[i, ax, f] = 0.0
        c[j, :, f] = c[i, :, f]                                  # (a trapped pair is a contact at half a box length only through the image)
        c[j, ax, f] = -d
    b[:, 11] = 0.0
    c[n1 + 5, 0, 12] = np.inf
    c[n1 + 9, 1, 13] = np.nan
    ch = np.ones(N, np.uint32); ch[n1:] = 2
    if ids == "chains":
        ch = rng.integers(0, 3, N).astype(np.uint32)
    elif ids == "one":
        ch[:] = 0
    return c, b, ch, np.arange(n1, dtype=np.uint32), np.arange(n1, N, dtype=np.uint32)


def _trapped_threshold_cases(seed=5, n1=8, n2=64, F=64):
    """Frames whose pair (first atom 0, first second atom) sits at an image-integer trap, with the float32 threshold whose square is EXACTLY the
    oracle's d^2 of that pair (frames for which no float32 squares to it are left out): `d^2 <= thr^2` holds for the reference and fails for a
    kernel whose image integer gave a larger |d - b r| -- the contact list itself tells whether the trapped batch was redone pair by pair."""
    rng = np.random.default_rng(seed)
    N = n1 + n2
    c = rng.uniform(0, 10, size=(N, 3, F)).astype(np.float32)
    b = np.full((3, F), 40.0, np.float32)
    s1, s2 = np.arange(n1, dtype=np.uint32), np.arange(n1, N, dtype=np.uint32)
    ch = np.ones(N, np.uint32); ch[n1:] = 2
    for f, (bl, d) in enumerate(_image_integer_traps(rng, F)):
        b[0, f] = bl
        c[0, 0, f] = 0.0
        c[n1, :, f] = c[0, :, f]
        c[n1, 0, f] = -d
    d2 = oracle.dist_trajectory(c, b, s1, s2, ch, False, True, squared=True)
    cases = []
    for f in range(F):
        thr = np.float32(np.sqrt(np.float64(d2[f, 0])))
        if np.float32(thr * thr) == d2[f, 0]:
            cases.append((c[:, :, f:f + 1].copy(), b[:, f:f + 1].copy(), float(thr), _contact_lists(d2[f:f + 1], s1, s2, thr)))
    assert len(cases) >= 5
    return cases, ch, s1, s2


@pytest.mark.parametrize("ids", ["selections", "chains", "one"])
def test_rectangular_contact_kernel_is_the_reference_order_and_image(ids):
    """k_contacts_count_rect / k_contacts_fill_rect (round 6: second atoms in registers, row tiles, packed arithmetic): rows that end inside a
    run of 16 and inside a tile of 64 (n2 = 70, 150), first atoms not a multiple of 8, frames not a multiple of 64, several chunks of
    frames, traps, a zero box, inf / NaN.  Against the oracle's squared distances in the reference's (frame, i, j) order, and equal to the
    pair-table walk's lists."""
    for n2 in (70, 150):
        c, b, ch, s1, s2 = _rect_contact_case(ids, n2)
        with np.errstate(all="ignore"):
            d2 = oracle.dist_trajectory(c, b, s1, s2, ch, False, True, squared=True)
            # a threshold around half a box length catches the trapped pairs (their image decides), 6 A the ordinary ones
            for thr in (6.0, 21.5):
                want = _contact_lists(d2, s1, s2, thr)
                assert sum(len(x) for x in want) > 0
                for budget, sink in ((256 << 20, False), (12 * 64 * 8, True)):
                    got = E.contacts_trajectory(c, b, s1, s2, ch, False, True, thr, budget_bytes=budget, device_sink=sink)
                    assert got == want, (n2, thr, budget)
                assert E.contacts_trajectory(c, b, s1, s2, ch, False, True, thr, avoid=1) == want
                for ni in (5, 8, 32):                                # rows per group (a large call gets up to 32; 5: the last group is short)
                    assert E.contacts_trajectory(c, b, s1, s2, ch, False, True, thr, avoid=ni << 8, budget_bytes=12 * 64 * 8 if ni == 5 else 256 << 20) == want, ni
            # pbc = False: no image at all
            d2o = oracle.dist_trajectory(c, b, s1, s2, ch, False, False, squared=True)
            assert E.contacts_trajectory(c, b, s1, s2, ch, False, False, 6.0) == _contact_lists(d2o, s1, s2, 6.0)


def test_rectangular_contact_kernel_trapped_pairs_decide_contacts():
    """A threshold whose square is exactly the reference's d^2 of a trapped pair: the pair is a contact only if the batch it sits in was
    redone with the correctly rounded divisions (mutation-checked: with the accumulated test switched off this fails)."""
    cases, ch, s1, s2 = _trapped_threshold_cases()
    for c, b, thr, want in cases:
        assert E.contacts_trajectory(c, b, s1, s2, ch, False, True, thr) == want


def test_rectangular_contact_kernel_with_more_second_atoms_than_the_fill_pass_stages():
    """n2 = 4 200 (> 4 096: the fill pass reads the second atoms from memory instead of its LDS copy), three first atoms, five frames."""
    rng = np.random.default_rng(8)
    n1, n2, F = 3, 4200, 5
    N = n1 + n2
    c = rng.uniform(0, 30, size=(N, 3, F)).astype(np.float32)
    b = np.full((3, F), 30.0, np.float32)
    ch = np.ones(N, np.uint32); ch[n1:] = 2
    s1, s2 = np.arange(n1, dtype=np.uint32), rng.permutation(np.arange(n1, N)).astype(np.uint32)
    d2 = oracle.dist_trajectory(c, b, s1, s2, ch, False, True, squared=True)
    want = _contact_lists(d2, s1, s2, 5.0)
    assert sum(len(x) for x in want) > 100
    assert E.contacts_trajectory(c, b, s1, s2, ch, False, True, 5.0) == want
    assert E.contacts_trajectory(c, b, s1, s2, ch, False, True, 5.0, avoid=2 << 8) == want


def test_selfdist_contacts_through_the_rectangular_kernels_unequal_selections():
    """contacts_trajectory(selfdist=True) pairs (sel1[i], sel2[j]) for j > i (distance_utils.pyx:76) -- also when the two selections differ in
    length and content; rows around the diagonal end inside runs of 16 and tiles of 64; every group size; equal to the pair-table walk."""
    rng = np.random.default_rng(41)
    N, F = 260, 70
    c = rng.uniform(0, 25, size=(N, 3, F)).astype(np.float32)
    b = np.full((3, F), 25.0, np.float32)
    ch = rng.integers(0, 3, N).astype(np.uint32)
    for n1, n2 in ((200, 200), (70, 150), (150, 70), (33, 64), (1, 40)):
        s1 = rng.permutation(N)[:n1].astype(np.uint32); s2 = rng.permutation(N)[:n2].astype(np.uint32)
        d2 = oracle.dist_trajectory(c, b, s1, s2, ch, True, True, squared=True)
        pairs = [(i, j) for i in range(n1) for j in range(i + 1, n2)]
        assert d2.shape[1] == len(pairs)
        want = []
        for f in range(F):
            hit = np.nonzero(d2[f] <= np.float32(36.0))[0]
            want.append([int(v) for k in hit for v in (s1[pairs[k][0]], s2[pairs[k][1]])])
        assert sum(len(x) for x in want) > 0 or n1 == 1
        for avoid in (1, 0, 5 << 8, 32 << 8):
            assert E.contacts_trajectory(c, b, s1, s2, ch, True, True, 6.0, avoid=avoid, budget_bytes=12 * 64 * 8 if avoid == 0 else 256 << 20) == want, (n1, n2, avoid)


@pytest.mark.parametrize("ids", ["selections", "chains", "one"])
def test_contact_lists_of_few_frames_take_lanes_along_the_second_atoms(ids):
    """Calls of at most 16 frames (get_collisions has one) count with lanes along the SECOND ATOMS (k_contacts_count_rect_few, round 6): slices
    of the rectangular case -- one frame, the frames with the zero box / inf / NaN coordinates, sixteen frames with their image-integer traps --
    against the oracle and equal to the lanes-along-frames kernel; selfdist too; rows per group 1 / 5 / 32."""
    for n2 in (70, 150):
        c, b, ch, s1, s2 = _rect_contact_case(ids, n2)
        for lo, hi in ((0, 1), (10, 14), (0, 16), (54, 70)):
            cs, bs = np.ascontiguousarray(c[:, :, lo:hi]), np.ascontiguousarray(b[:, lo:hi])
            with np.errstate(all="ignore"):
                for pbc in (True, False):
                    d2 = oracle.dist_trajectory(cs, bs, s1, s2, ch, False, pbc, squared=True)
                    for thr in (6.0, 21.5):
                        want = _contact_lists(d2, s1, s2, thr)
                        for avoid in (0, 2, 5 << 8, 32 << 8):            # few-frames kernel; lanes along frames; groups of 5 and 32 rows
                            assert E.contacts_trajectory(cs, bs, s1, s2, ch, False, pbc, thr, avoid=avoid) == want, (n2, lo, hi, pbc, thr, avoid)
    # selfdist, unequal selections, one and three frames
    rng = np.random.default_rng(43)
    N = 200
    c = rng.uniform(0, 20, size=(N, 3, 3)).astype(np.float32)
    b = np.full((3, 3), 20.0, np.float32)
    ch = rng.integers(0, 3, N).astype(np.uint32)
    for n1, n2 in ((150, 150), (70, 130), (130, 70)):
        s1 = rng.permutation(N)[:n1].astype(np.uint32); s2 = rng.permutation(N)[:n2].astype(np.uint32)
        pairs = [(i, j) for i in range(n1) for j in range(i + 1, n2)]
        for F in (1, 3):
            d2 = oracle.dist_trajectory(c[:, :, :F].copy(), b[:, :F].copy(), s1, s2, ch, True, True, squared=True)
            want = [[int(v) for k in np.nonzero(d2[f] <= np.float32(25.0))[0] for v in (s1[pairs[k][0]], s2[pairs[k][1]])] for f in range(F)]
            for avoid in (0, 2, 1):
                assert E.contacts_trajectory(c[:, :, :F].copy(), b[:, :F].copy(), s1, s2, ch, True, True, 5.0, avoid=avoid) == want, (n1, n2, F, avoid)


def test_host_calls_pack_the_selected_atoms_before_the_upload():
    """csrc/host_pack.h (round 6): the host entry points upload only the rows of the atoms a call selects when those are at most a quarter
    of a (> 1 MB) coordinate array -- unsorted selections with repeats, the packed numbering, the coordinates' rows, the way back for
    contact lists; and the cases that upload the array as it is (small arrays, selections of more than a quarter of the atoms)."""
    rng = np.random.default_rng(17)
    N, F = 3000, 40                                              # 1.44 MB
    c = rng.normal(size=(N, 3, F)).astype(np.float32)
    s1 = rng.integers(0, N, size=200).astype(np.uint32); s2 = np.array([7, 7, 2999, 0, 1500, 7], np.uint32)
    on, uniq, packed, remap, back = E.pack_atoms(c, [s1, s2])
    flat = np.concatenate([s1, s2])
    assert on and np.array_equal(uniq, np.unique(flat)) and len(uniq) * 4 <= N
    assert np.array_equal(packed, c[uniq.astype(np.int64)])                 # the rows, in the packed order
    assert np.array_equal(uniq[remap.astype(np.int64)], flat)               # the selection names the same atoms
    assert np.array_equal(back, flat)                                       # and packed indices translate back (contact lists)
    on, uniq, _, _, _ = E.pack_atoms(c, [rng.permutation(N)[:800].astype(np.uint32)])
    assert not on and len(uniq) == 800                                      # more than a quarter of the atoms: as it is
    on, _, _, _, _ = E.pack_atoms(c[:, :, :20].copy(), [s1])
    assert not on                                                           # an array of less than a megabyte: as it is
    on, uniq, packed, _, _ = E.pack_atoms(c, [np.array([5], np.uint32)])
    assert on and uniq.tolist() == [5] and np.array_equal(packed[0], c[5])


def test_selfdist_of_few_frames_goes_through_the_triangular_row_kernel():
    """Round 6 (late): a selfdist call of large selections (>= 700 atoms up to 32 frames, >= 1 500 at any frame count) takes the row kernel's triangular
    form (k_dist_rows<.., TRI>: lanes along the second atoms, tasks below the diagonal skipped, the reference's condensed order written
    directly) -- the pair-table kernel runs its lanes along frames.  Equal and UNEQUAL selections (pairs i < j over sel1[i], sel2[j]: distance_utils.pyx:140-150), one and
    three frames, periodic with mixed chains and open, squared; the oracle's bits, and the pair-table kernel's (avoid bit 64) on the same call."""
    rng = np.random.default_rng(91)
    N = 500
    ch = rng.integers(0, 3, size=N).astype(np.uint32)
    sa = rng.permutation(N)[:250].astype(np.uint32)              # rows of four atoms per lane, 16-byte stores
    sb = rng.permutation(N)[:190].astype(np.uint32)              # one atom per lane
    sc = rng.permutation(N)[:330].astype(np.uint32)              # two
    for F in (1, 3):
        c = rng.uniform(-25, 25, size=(N, 3, F)).astype(np.float32)
        b = rng.uniform(18, 30, size=(3, F)).astype(np.float32)
        for a1, a2 in ((sa, sa), (sa, sb), (sb, sa), (sc, sc), (sb, sc)):
            for pbc in (True, False):
                want = oracle.dist_trajectory(c, b, a1, a2, ch, True, pbc)
                # (the library's own choice takes the triangular form from 700 atoms on: avoid bit 16 = the row kernel wherever it applies)
                assert np.array_equal(E.dist_trajectory(c, b, a1, a2, ch, True, pbc, avoid=16), want), (F, len(a1), len(a2), pbc)
                assert np.array_equal(E.dist_trajectory(c, b, a1, a2, ch, True, pbc, avoid=16 | 8), want)      # (without the 16-byte stores)
                assert np.array_equal(E.dist_trajectory(c, b, a1, a2, ch, True, pbc, avoid=64), want)
                assert np.array_equal(E.dist_trajectory(c, b, a1, a2, ch, True, pbc), want)
        want = oracle.dist_trajectory(c, b, sa, sa, ch, True, True, squared=True)
        assert np.array_equal(E.dist_trajectory(c, b, sa, sa, ch, True, True, squared=True, avoid=16), want)
    # ... and at the library's own choice: 720 atoms of one frame
    big = rng.permutation(N * 2)[:720].astype(np.uint32)
    c = rng.uniform(-25, 25, size=(2 * N, 3, 1)).astype(np.float32)
    b = rng.uniform(18, 30, size=(3, 1)).astype(np.float32)
    ch2 = rng.integers(0, 3, size=2 * N).astype(np.uint32)
    assert np.array_equal(E.dist_trajectory(c, b, big, big, ch2, True, True), oracle.dist_trajectory(c, b, big, big, ch2, True, True))


def test_which_kernel_dist_trajectory_takes_at_few_frames_and_for_large_selfdist_calls():
    """The choice rule of run_dist_trajectory (round 6, measured: profiles/r6_dist_few_frames_probe.txt, r6_dist_self_tri_probe.txt), pinned on the CPU tier
    through the emulator's backend: rectangular calls of <= 32 frames take the row kernel wherever it applies (33 frames of a small result: the tile
    kernel); selfdist takes the row kernel's triangular form from 700 atoms up to 32 frames and from 1 500 atoms at any frame count, else the pair table;
    rows too short for the row kernel: the block-per-frame kernel.  (One frame each of tiny coordinates: the choice depends on the shape alone.)"""
    rng = np.random.default_rng(3)
    N = 1600
    ch = np.zeros(N, np.uint32)
    sel = np.arange(N, dtype=np.uint32)

    def chosen(n1, n2, F, selfd, pbc=False, avoid=0):
        c = rng.uniform(0, 9, size=(N, 3, F)).astype(np.float32)
        b = np.full((3, F), 30.0, np.float32)
        E.dist_trajectory(c, b, sel[:n1], sel[:n2], ch, selfd, pbc, avoid=avoid)
        return E.last_dist_kernel()

    assert "k_dist_rows<false, 1, false>" in chosen(70, 60, 1, False)
    assert "k_dist_rows<false, 1, false>" in chosen(70, 60, 32, False)
    assert "k_dist_rect" in chosen(70, 60, 33, False)
    assert "k_dist_frame" in chosen(70, 30, 1, False)
    assert "k_dist_pairs" in chosen(450, 450, 1, True)
    assert "k_dist_rows<false, 4, true, true>" in chosen(720, 720, 1, True)
    assert "k_dist_rows<true, 4, true, true>" in chosen(720, 720, 2, True, pbc=True)
    assert "k_dist_pairs" in chosen(720, 720, 1, True, avoid=64)
    assert "k_dist_pairs" in chosen(720, 720, 33, True)
    assert "k_dist_rows<false, 4, true, true>" in chosen(1536, 1536, 1, True)


def test_short_row_calls_of_few_frames_take_the_row_kernel_the_other_way_round():
    """Round 6 (late): a rectangular call of few frames whose rows are too short for the row kernel and whose FIRST selection is long (a receptor's atoms
    x a ligand's, beyond what the block-per-frame kernel stages) runs the row kernel with the selections swapped -- lanes along the first selection,
    transposed stores; every operation on the separation is odd, so the swapped pair has the reference's bits.  One and three frames, periodic with
    mixed chains and open, squared, a zero box and a NaN coordinate; the oracle's bits, the tile kernel's (avoid bit 128) and the kernel's name."""
    rng = np.random.default_rng(23)
    N = 4400
    ch = rng.integers(0, 3, size=N).astype(np.uint32)
    s1 = rng.permutation(N)[:4200].astype(np.uint32)
    s2 = rng.permutation(N)[:30].astype(np.uint32)
    for F in (1, 3):
        c = rng.uniform(-25, 25, size=(N, 3, F)).astype(np.float32)
        b = rng.uniform(18, 30, size=(3, F)).astype(np.float32)
        if F == 3:
            b[:, 1] = 0.0
            c[int(s1[5]), 1, 2] = np.nan
        for pbc in (True, False):
            with np.errstate(all="ignore"):
                want = oracle.dist_trajectory(c, b, s1, s2, ch, False, pbc)
            assert np.array_equal(E.dist_trajectory(c, b, s1, s2, ch, False, pbc), want, equal_nan=True), (F, pbc)
            assert ", false, true>" in E.last_dist_kernel(), E.last_dist_kernel()
            assert np.array_equal(E.dist_trajectory(c, b, s1, s2, ch, False, pbc, avoid=128), want, equal_nan=True)
            assert "k_dist_rect" in E.last_dist_kernel()
        with np.errstate(all="ignore"):
            want = oracle.dist_trajectory(c, b, s1, s2, ch, False, True, squared=True)
        assert np.array_equal(E.dist_trajectory(c, b, s1, s2, ch, False, True, squared=True), want, equal_nan=True)
