"""GPU tier (-m gpu), distance_utils row: every function through the C ABI on the MI355X, bit-exact against the
real reference's outputs (tests/golden/distance_cases.npz) and against the oracle at larger sizes."""
import numpy as np
import pytest

from oracle import oracle
from tests.cases import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return golden("distance_cases.npz")


def test_dist_trajectory_bit_exact(g):
    from moleculekit_amd.distance_utils import dist_trajectory
    c, b, ch = g["coords"], g["box"], g["chains"]
    for pbc in (0, 1):
        r = np.zeros_like(g[f"dist_cross_pbc{pbc}"])
        assert dist_trajectory(c, b, g["sel1"], g["sel2"], ch, False, bool(pbc), r) is None
        assert np.array_equal(r, g[f"dist_cross_pbc{pbc}"])
        r = np.zeros_like(g[f"dist_self_pbc{pbc}"])
        dist_trajectory(c, b, g["sel2"], g["sel2"], ch, True, bool(pbc), r)
        assert np.array_equal(r, g[f"dist_self_pbc{pbc}"])


def test_contacts_and_collisions_match_reference_lists(g):
    from moleculekit_amd.distance_utils import contacts_trajectory, get_collisions
    c, b, ch = g["coords"], g["box"], g["chains"]
    res = contacts_trajectory(c, b, g["sel1"], g["sel2"], ch, False, True, 12.0)
    assert np.array_equal([len(x) // 2 for x in res], g["contacts_counts"])
    assert np.array_equal(np.concatenate([np.asarray(x, np.int64) for x in res]), g["contacts_flat"])
    res = contacts_trajectory(c, b, g["sel2"], g["sel2"], ch, True, False, 15.0)
    assert np.array_equal([len(x) // 2 for x in res], g["contacts_self_counts"])
    assert np.array_equal(np.concatenate([np.asarray(x, np.int64) for x in res]), g["contacts_self_flat"])
    col = get_collisions(np.ascontiguousarray(c[:20, :, 0]), np.ascontiguousarray(c[20:50, :, 0]), 14.0)
    assert np.array_equal(col, g["collisions"])


def test_reductions_bit_exact(g):
    from moleculekit_amd.distance_utils import dist_trajectory_reduction, dist_trajectory_reduction_pairs
    c, b, m = g["coords"], g["box"], g["masses"]
    g1 = [list(x) for x in g["groups1"]]; g2 = [list(x) for x in g["groups2"]]
    for r1 in (0, 1):
        for r2 in (0, 1):
            for pbc in (0, 1):
                r = np.zeros_like(g[f"red_{r1}{r2}_pbc{pbc}"])
                dist_trajectory_reduction(c, b, g1, g2, g["gchains1"], g["gchains2"], False, bool(pbc), m, r1, r2, r)
                assert np.array_equal(r, g[f"red_{r1}{r2}_pbc{pbc}"]), (r1, r2, pbc)
    r = np.zeros_like(g["red_self"])
    dist_trajectory_reduction(c, b, g2, g2, g["gchains2"], g["gchains2"], True, True, m, 0, 0, r)
    assert np.array_equal(r, g["red_self"])
    r = np.zeros_like(g["red_pairs"])
    dist_trajectory_reduction_pairs(c, b, g1, g2[:5], g["gchains1"], g["gchains2"][:5], True, m, 0, 1, r)
    assert np.array_equal(r, g["red_pairs"])


def test_cdist_pdist_squareform_like_reference_tests(g):
    """tests/test_distance.py:1-28 (known answers) + bit-exact golden in 1, 2, 3 and 5 dimensions."""
    from moleculekit_amd.distance import cdist, pdist, squareform
    x = np.array([0, 1, 2])[:, None]; y = np.array([3, 4, 5])[:, None]
    assert np.allclose(cdist(x, y), [[3.0, 4.0, 5.0], [2.0, 3.0, 4.0], [1.0, 2.0, 3.0]])
    assert np.allclose(cdist(np.array([[0, 1], [2, 3]]), np.array([[4, 5], [6, 7], [8, 9]])),
                       [[5.656854, 8.485281, 11.313708], [2.828427, 5.656854, 8.485281]])
    assert np.allclose(pdist(np.array([[4, 5], [6, 7], [8, 9]])), [2.828427, 5.656854, 2.828427])
    for D in (1, 2, 3, 5):
        assert np.array_equal(cdist(g[f"cdist_a{D}"], g[f"cdist_b{D}"]), g[f"cdist_r{D}"])
        assert np.array_equal(pdist(g[f"cdist_b{D}"]), g[f"pdist_r{D}"])
    assert np.array_equal(squareform(g["pdist_r3"]), g["squareform"])


def test_trajectory_scale_against_oracle():
    """A cfg4-flavoured trajectory slice: 2000 atoms x 300 frames, 120 x 150 atom pairs, periodic by chain;
    tile edges in both directions; bit-exact against the oracle."""
    from moleculekit_amd.distance_utils import dist_trajectory
    rng = np.random.default_rng(17)
    N, F = 2000, 300
    c = rng.uniform(0, 66.9, size=(N, 3, F)).astype(np.float32)
    b = np.full((3, F), 66.9, np.float32)
    ch = (np.arange(N) // 250).astype(np.uint32)
    s1 = np.sort(rng.choice(N, 120, replace=False)).astype(np.uint32)
    s2 = np.sort(rng.choice(N, 150, replace=False)).astype(np.uint32)
    r = np.zeros((F, 120 * 150), np.float32)
    dist_trajectory(c, b, s1, s2, ch, False, True, r)
    assert np.array_equal(r, oracle.dist_trajectory(c, b, s1, s2, ch, False, True))
    diff = ch[s1][:, None] != ch[s2][None, :]                 # minimum image only across chains (:49)
    assert r.reshape(F, 120, 150)[:, diff].max() <= np.sqrt(3) * 66.9 / 2 + 1e-3
    r2 = np.zeros((F, 150 * 149 // 2), np.float32)
    dist_trajectory(c, b, s2, s2, ch, True, True, r2)
    assert np.array_equal(r2, oracle.dist_trajectory(c, b, s2, s2, ch, True, True))


def test_ragged_runs_zero_box_and_contact_lists_against_oracle():
    """What the pair-run traversal has to get right on the hardware: runs of 16 pairs whose first atom changes inside a
    batch of four (n2 = 5, self pairs), pair and frame counts that are not multiples of 64 or 16, a frame with a zero box
    (pbc -> NaN like the reference: the image-shift test must send it to the division), and the same traversal under the
    contact lists (reference order: frame, i, j; distance_utils.pyx:59-93)."""
    from moleculekit_amd.distance_utils import dist_trajectory, contacts_trajectory
    rng = np.random.default_rng(3)
    N, F = 150, 70
    c = rng.uniform(-30, 30, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(15, 25, size=(3, F)).astype(np.float32)
    b[:, 5] = 0.0
    ch = rng.integers(0, 3, size=N).astype(np.uint32)
    s1 = np.arange(0, 90, dtype=np.uint32); s2 = np.arange(40, 150, dtype=np.uint32)
    thr = 12.0
    for selfd, a, bb in ((False, s1, s2), (True, s2, s2), (True, s1[:7], s2[:20]), (False, s1[:9], s2[:5]), (False, s1[:1], s2[:3])):
        exp = oracle.dist_trajectory(c, b, a, bb, ch, selfd, True)
        got = np.zeros_like(exp)
        dist_trajectory(c, b, a, bb, ch, selfd, True, got)
        assert np.array_equal(got, exp, equal_nan=True)
        assert np.isnan(exp[5]).any()
        d2 = oracle.dist_trajectory(c, b, a, bb, ch, selfd, True, squared=True)
        if selfd:
            table = [(a[i], bb[j]) for i in range(len(a)) for j in range(i + 1, len(bb))]
        else:
            table = [(a[i], bb[j]) for i in range(len(a)) for j in range(len(bb))]
        lists = contacts_trajectory(c, b, a, bb, ch, selfd, True, thr)
        assert len(lists) == F
        for f in range(F):
            hits = np.nonzero(d2[f] <= np.float32(thr) * np.float32(thr))[0]          # NaN: no contact (:82)
            assert lists[f] == [int(v) for k in hits for v in table[k]]


def test_analytic_two_angstrom_box_case():
    """The analytic 2 A-box case in the spirit of tests/test_metricdistance.py:99-135, through the distance_utils functions themselves
    with the arguments the reference's drivers build (projections/util.py:24-70: chains digitized by chain or by selection, a zero box
    and pbc = False for periodic=None): wrapped and unwrapped distances, the closest-atom and centre-of-mass reductions, contact lists."""
    from moleculekit_amd import distance_utils as du
    coords = np.zeros((4, 3, 2), np.float32)
    coords[1, 0, :] = 1.5; coords[2, 1, :] = 0.5; coords[3, 2, 1] = 1.75
    box = np.full((3, 2), 2.0, np.float32)
    by_chain = np.array([0, 1, 1, 2], np.uint32)                       # chains A, B, B, C
    s1, s2 = np.array([0], np.uint32), np.array([1, 2, 3], np.uint32)
    d = np.zeros((2, 3), np.float32)
    du.dist_trajectory(coords, np.zeros_like(box), s1, s2, np.zeros(4, np.uint32), False, False, d)           # periodic=None
    assert np.allclose(d, [[1.5, 0.5, 0.0], [1.5, 0.5, 1.75]])
    du.dist_trajectory(coords, box, s1, s2, by_chain, False, True, d)                                         # periodic="chains"
    assert np.allclose(d, [[0.5, 0.5, 0.0], [0.5, 0.5, 0.25]])
    by_sel = np.array([1, 2, 2, 2], np.uint32)                          # periodic="selections"
    du.dist_trajectory(coords, box, s1, s2, by_sel, False, True, d)
    assert (d <= 0.4).tolist() == [[False, False, True], [False, False, True]]
    masses = np.array([12.011, 14.0067, 15.9994, 12.011], np.float32)
    red = np.zeros((2, 2), np.float32)
    du.dist_trajectory_reduction(coords, box, [[0]], [[1, 2], [3]], np.array([0], np.uint32), np.array([1, 2], np.uint32), False, True, masses, 0, 0, red)
    assert np.allclose(red, [[0.5, 0.0], [0.5, 0.25]])
    com = np.zeros((2, 1), np.float32)
    du.dist_trajectory_reduction(coords, np.zeros_like(box), [[0]], [[1, 2]], np.zeros(1, np.uint32), np.zeros(1, np.uint32), False, False, masses, 0, 1, com)
    w = masses[1:3]
    cx, cy = 1.5 * w[0] / w.sum(), 0.5 * w[1] / w.sum()
    assert np.allclose(com, np.sqrt(cx * cx + cy * cy), atol=1e-6)
    con = du.contacts_trajectory(coords, box, s1, s2, by_chain, False, True, 0.3)
    assert con == [[0, 3], [0, 3]]


def test_image_shift_at_the_half_box_boundary_is_bit_exact():
    """Separations on / one ulp either side of half a box edge: the image shift round(d / box) must see the
    correctly rounded quotient (a reciprocal-multiply shortcut would flip some of these)."""
    from moleculekit_amd.distance_utils import dist_trajectory
    from tests.test_distance_cpu import _half_box_case
    c, b, ch, s1, s2 = _half_box_case()
    r = np.zeros((c.shape[2], len(s2)), np.float32)
    dist_trajectory(c, b, s1, s2, ch, False, True, r)
    assert np.array_equal(r, oracle.dist_trajectory(c, b, s1, s2, ch, False, True))


def test_more_than_65535_rows_and_groups():
    """Launch geometry: the row / group dimension is a grid-stride loop, not gridDim.y (capped at 65535).  cdist with
    70 000 rows, dist_trajectory with a 70 000-atom first selection, contact lists over the same, centres of mass of
    70 000 one-atom groups -- each against numpy / the oracle, bit for bit."""
    from moleculekit_amd.distance_utils import cdist, contacts_trajectory, dist_trajectory, dist_trajectory_reduction
    rng = np.random.default_rng(65)
    n1, n2, F = 70000, 3, 2
    xyz = rng.uniform(-20, 20, size=(n1 + n2, 3, F)).astype(np.float32)
    box = np.full((3, F), 31.0, np.float32)
    chains = (np.arange(n1 + n2) % 7).astype(np.uint32)
    sel1, sel2 = np.arange(n1, dtype=np.uint32), np.arange(n1, n1 + n2, dtype=np.uint32)
    # cdist: rows beyond 65535 are reached
    a, b = np.ascontiguousarray(xyz[:n1, :, 0]), np.ascontiguousarray(xyz[n1:, :, 0])
    got = np.zeros((n1, n2), np.float32)
    cdist(a, b, got)
    d = a[:, None, :] - b[None, :, :]
    want = np.sqrt(((d[..., 0] * d[..., 0]) + (d[..., 1] * d[..., 1])) + (d[..., 2] * d[..., 2]))   # float32, the reference's order
    assert np.array_equal(got, want.astype(np.float32))
    # dist_trajectory + contacts on the same selections vs the oracle
    got = np.zeros((F, n1 * n2), np.float32)
    dist_trajectory(xyz, box, sel1, sel2, chains, False, True, got)
    want = oracle.dist_trajectory(xyz, box, sel1, sel2, chains, False, True)
    assert np.array_equal(got, want)
    res = contacts_trajectory(xyz, box, sel1, sel2, chains, False, True, 6.0)
    d2 = oracle.dist_trajectory(xyz, box, sel1, sel2, chains, False, True, squared=True)
    thr = np.float32(6.0) * np.float32(6.0)
    for f in range(F):
        flat = np.asarray(res[f], np.int64).reshape(-1, 2)
        assert flat[:, 0].max() > 65535
        assert np.array_equal(flat[:, 0] * n2 + (flat[:, 1] - n1), np.nonzero(d2[f] <= thr)[0])   # same pairs, the (i, j) order
    # 70 000 one-atom groups through the centre-of-mass reduction: com == the atom, distances == dist_trajectory
    m = rng.uniform(1, 16, size=n1 + n2).astype(np.float32)
    got = np.zeros((F, n1 * n2), np.float32)
    dist_trajectory_reduction(xyz, box, [[i] for i in range(n1)], [[n1 + j] for j in range(n2)], chains[:n1].copy(),
                              chains[n1:].copy(), False, True, np.ones_like(m), 1, 1, got)
    assert np.array_equal(got, want)


def test_rectangular_and_pair_table_kernels_over_many_tiles_match_oracle():
    """Round 4: dist_trajectory without selfdist runs k_dist_rect (second atoms in registers, 8 first atoms per block), with
    selfdist the pair-table kernel; both deal their tiles to the XCDs in contiguous ranges of a 1-D grid padded to a multiple
    of 8.  Shapes with hundreds of tiles whose count is NOT a multiple of 8, ragged edges on every axis (frames, first and
    second atoms), rows that do and do not start on 16 bytes (16-byte / 4-byte store paths), pbc on and off, squared and
    not -- against the oracle, bit for bit."""
    from moleculekit_amd.distance_utils import dist_trajectory
    rng = np.random.default_rng(23)
    N, F = 900, 203                                            # 4 frame slabs, the last one ragged
    c = rng.uniform(-40, 40, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(30, 45, size=(3, F)).astype(np.float32)
    ch = rng.integers(0, 5, size=N).astype(np.uint32)
    for n1, n2 in ((19, 132), (8, 64), (37, 203), (1, 500)):
        s1 = rng.choice(N, n1, replace=False).astype(np.uint32)
        s2 = rng.choice(N, n2, replace=False).astype(np.uint32)
        for pbc in (False, True):
            r = np.full((F, n1 * n2), -3.0, np.float32)
            dist_trajectory(c, b, s1, s2, ch, False, pbc, r)
            assert np.array_equal(r, oracle.dist_trajectory(c, b, s1, s2, ch, False, pbc)), (n1, n2, pbc)
    s = rng.choice(N, 61, replace=False).astype(np.uint32)    # 1 830 pairs: 29 pair tiles x 4 slabs = 116 tiles
    for pbc in (False, True):
        r = np.full((F, 61 * 60 // 2), -3.0, np.float32)
        dist_trajectory(c, b, s, s, ch, True, pbc, r)
        assert np.array_equal(r, oracle.dist_trajectory(c, b, s, s, ch, True, pbc)), pbc


def test_row_kernel_over_many_waves_matches_oracle():
    """Round 4: rectangular calls with rows of >= 64 second atoms take k_sel_to_frames + k_dist_rows (selections turned
    frame-major, a wave per frame writes whole pieces of a row): shapes for 1, 2 and 4 second atoms per lane, ragged on every
    axis (frames not a multiple of the turning kernel's 64, first atoms not a multiple of a wave's 16, second atoms not a
    multiple of 64), wave-task counts that are not multiples of 4 or 32, unsorted and repeated atoms, a zero box edge, pbc on
    and off (squared distances: the CPU tier's emulated run); and the bench leg's shape (200 x 500) on a slice of frames -- against the oracle, bit for bit,
    with a sentinel in every element first."""
    from moleculekit_amd.distance_utils import dist_trajectory
    rng = np.random.default_rng(29)
    N, F = 900, 203
    c = rng.uniform(-40, 40, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(30, 45, size=(3, F)).astype(np.float32)
    b[2, 77] = 0.0
    ch = rng.integers(0, 5, size=N).astype(np.uint32)
    for n1, n2 in ((40, 52), (41, 64), (24, 128), (20, 256), (30, 500), (50, 260), (200, 500), (33, 1000)):
        s1 = rng.integers(0, N, size=n1).astype(np.uint32)
        s2 = rng.integers(0, N, size=n2).astype(np.uint32)
        for pbc in (False, True):
            r = np.full((F, n1 * n2), -3.0, np.float32)
            dist_trajectory(c, b, s1, s2, ch, False, pbc, r)
            exp = oracle.dist_trajectory(c, b, s1, s2, ch, False, pbc)
            assert np.array_equal(r, exp, equal_nan=True), (n1, n2, pbc)


def test_block_per_frame_kernel_over_many_blocks_and_every_kernel_on_the_same_shapes():
    """Round 5: k_dist_frame (a block per frame / slice stages both selections in LDS and walks the pair list in memory order) at
    sizes where its bookkeeping matters -- thousands of blocks whose count is not a multiple of 8, several slices per frame, short
    rectangular rows (MetricDistance's protein x ligand shape), frame rows that do not start on 16 bytes, more than 1 024 atoms (the
    big LDS tier), a zero box edge; triangular lists (MetricSelfDistance's shape: the pair-table kernel) beside them -- and every
    kernel that can take a shape on that shape (``Context.set_dist_kernels``): all against the oracle, bit for bit,
    with a sentinel in every element first."""
    from moleculekit_amd import _lib
    from moleculekit_amd.distance_utils import dist_trajectory
    ctx = _lib.default_context()
    rng = np.random.default_rng(31)
    N, F = 2500, 203
    c = rng.uniform(-40, 40, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(30, 45, size=(3, F)).astype(np.float32)
    b[2, 77] = 0.0
    ch = rng.integers(0, 5, size=N).astype(np.uint32)
    shapes = [(False, 300, 30), (False, 30, 300), (False, 41, 101), (False, 7, 3), (False, 5, 1031), (True, 61, 61), (True, 450, 450), (True, 12, 90),
              (True, 90, 12), (True, 2, 2), (False, 20, 1300)]
    seen = set()
    try:
        for selfd, n1, n2 in shapes:
            s2 = rng.choice(N, n2, replace=False).astype(np.uint32)
            s1 = s2[:n1].copy() if (selfd and n1 <= n2) else rng.choice(N, n1, replace=False).astype(np.uint32)
            for pbc in (False, True):
                exp = oracle.dist_trajectory(c, b, s1, s2, ch, selfd, pbc)
                for avoid in (0, 1, 2, 3):
                    ctx.set_dist_kernels(avoid)
                    r = np.full(exp.shape, -3.0, np.float32)
                    dist_trajectory(c, b, s1, s2, ch, selfd, pbc, r)
                    seen.add(ctx.last_dist_kernel().split("<")[0])
                    assert np.array_equal(r, exp, equal_nan=True), (selfd, n1, n2, pbc, avoid, ctx.last_dist_kernel())
        # a frame count that is a multiple of four: the frame kernel's 16-byte staging loads
        c4, b4 = np.ascontiguousarray(c[:, :, :200]), np.ascontiguousarray(b[:, :200])
        s1, s2 = rng.choice(N, 300, replace=False).astype(np.uint32), rng.choice(N, 30, replace=False).astype(np.uint32)
        ctx.set_dist_kernels(0)
        for pbc in (False, True):
            r = np.full((200, 9000), -3.0, np.float32)
            dist_trajectory(c4, b4, s1, s2, ch, False, pbc, r)
            assert "k_dist_frame" in ctx.last_dist_kernel() and ", 4," in ctx.last_dist_kernel()
            assert np.array_equal(r, oracle.dist_trajectory(c4, b4, s1, s2, ch, False, pbc), equal_nan=True), pbc
    finally:
        ctx.set_dist_kernels(0)
    assert {"mkamd::k_dist_frame", "mkamd::k_dist_rect", "mkamd::k_build_atom_pairs + mkamd::k_dist_pairs", "mkamd::k_sel_to_frames + mkamd::k_dist_rows"} <= seen, seen


def test_results_at_four_byte_alignment_take_the_sixteen_byte_stores():
    """Round 5: every kernel stores 16 bytes per lane at whatever alignment a result row has (rows of an odd pitch -- the triangular
    list of 450 atoms has 101 025 pairs --, a result pointer that is an offset view: 4-byte aligned only).  The same calls into an
    aligned buffer and into one that starts 4, 8 and 12 bytes behind a 16-byte boundary, with a sentinel in front of and behind the
    result: bit for bit the oracle's, nothing written outside."""
    import torch
    from moleculekit_amd import _lib
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(77)
    N, F = 1200, 130
    c = rng.uniform(-40, 40, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(30, 45, size=(3, F)).astype(np.float32)
    ch = rng.integers(0, 5, size=N).astype(np.uint32)
    dc, db = torch.as_tensor(c, device=dev), torch.as_tensor(b, device=dev)
    dch = torch.as_tensor(ch.astype(np.int32), device=dev)
    seen = set()
    for selfd, n1, n2 in ((False, 20, 254), (False, 20, 256), (True, 131, 131), (False, 300, 30), (False, 301, 31), (False, 70, 70)):
        s2 = rng.choice(N, n2, replace=False).astype(np.uint32)
        s1 = s2.copy() if selfd else rng.choice(N, n1, replace=False).astype(np.uint32)
        d1, d2 = torch.as_tensor(s1.astype(np.int32), device=dev), torch.as_tensor(s2.astype(np.int32), device=dev)
        for pbc in (False, True):
            exp = oracle.dist_trajectory(c, b, s1, s2, ch, selfd, pbc)
            P = exp.shape[1]
            for off in (0, 1, 2, 3):
                buf = torch.full((F * P + 8,), -3.0, device=dev, dtype=torch.float32)
                out = buf[4 + off: 4 + off + F * P]
                assert out.data_ptr() % 16 == (4 * off) % 16
                ctx.dist_trajectory_dev(dc.data_ptr(), F, db.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2, dch.data_ptr(), selfd, pbc, False, out.data_ptr())
                torch.cuda.synchronize(dev)
                seen.add(ctx.last_dist_kernel().split("<")[0])
                h = buf.cpu().numpy()
                assert np.array_equal(h[4 + off: 4 + off + F * P].reshape(F, P), exp, equal_nan=True), (selfd, n1, n2, pbc, off, ctx.last_dist_kernel())
                assert (h[:4 + off] == -3.0).all() and (h[4 + off + F * P:] == -3.0).all(), (selfd, n1, n2, pbc, off)
    assert {"mkamd::k_dist_frame", "mkamd::k_build_atom_pairs + mkamd::k_dist_pairs", "mkamd::k_sel_to_frames + mkamd::k_dist_rows"} <= seen, seen


def test_the_short_square_root_is_the_correctly_rounded_one_for_every_float():
    """Round 5: the kernels' root is one exact-residual correction of x * rsq(x) (csrc/mk_device.h, mk_fsqrt_rn_ordinary) -- 8 issue slots
    instead of the provable form's 12 in kernels bound by instruction issue.  Correct rounding is a property of this chip's
    v_rsq_f32: the library checks it on the device over all 1 879 048 192 floats in [2^-96, inf) against v_sqrt_f32 + Tuckerman's
    test (mkamd_selftest_sqrt); and a sample of the same range against the host's correctly rounded sqrtf through the product's
    own API (cdist of 1-D points: sqrt((x - 0)^2) = |x| would prove nothing; squared distances of 3-D points do)."""
    from moleculekit_amd import _lib
    from moleculekit_amd.distance_utils import cdist
    assert _lib.default_context().selftest_sqrt() == (0, 0)
    rng = np.random.default_rng(7)
    a = (rng.uniform(-1, 1, size=(3000, 3)) * np.exp(rng.uniform(-40, 40, size=(3000, 1)))).astype(np.float32)
    b = (rng.uniform(-1, 1, size=(700, 3)) * np.exp(rng.uniform(-40, 40, size=(700, 1)))).astype(np.float32)
    got = np.zeros((3000, 700), np.float32)
    cdist(a, b, got)
    d = a[:, None, :] - b[None, :, :]                                  # float32, one rounding per operation, the reference's order
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    assert d2.dtype == np.float32 and np.array_equal(got, np.sqrt(d2))


# ------------------------------------------------------------------------------------------------
# the reference's OWN answers on its real trajectory (tests/metricdistance_real.py): own XTC reader ->
# own kernels, the calls MetricDistance makes (arguments as the reference's drivers built them)
# ------------------------------------------------------------------------------------------------
def test_known_answers_of_the_references_metricdistance_tests_on_the_gpu():
    """tests/metricdistance_known.py (test_metricdistance.py:99-181, :213-229, :329-352, :355-464, :467-493): every call those tests'
    projections make into distance_utils, through this package's functions: bit for bit the compiled reference, and the numbers the
    reference's tests assert (analytic molecules, 3PTB's 8.978174 / 3.8286476 / 2.8153415, pairs mode, periodic modes, truncate)."""
    from moleculekit_amd import distance_utils as du
    from tests import metricdistance_known as K, metricdistance_real as M
    g = K.load()
    traj = M.read_trajectory(M.load())
    with_answer = sum(K.verify(K.replay(du, g, key, traj), g, key) for key in map(str, g["names"]))
    assert len(g["names"]) == 22 and with_answer == 17


def test_reference_held_metricdistance_projections_on_the_gpu():
    from moleculekit_amd import distance_utils as du
    from tests import metricdistance_real as M
    g = M.load()
    coords, box = M.read_trajectory(g)
    report = {}
    for key in M.KEYS:
        res = M.run_projection(du, coords, box, g, key)
        exact, worst = M.check(res, g, key)                      # allclose(atol=1e-3) with the held arrays, as the reference asserts
        report[key] = (exact, worst)
        assert exact, f"{key}: not bit-exact with the compiled reference"
    print("metricdistance_real:", report)
    # ... and the device XTC decoder feeds the same bits to the same kernels (coords frame-major on the device -> [N,3,F])
    import torch
    from moleculekit_amd.xtc import read_xtc_frames_dev
    xyz, _, _, _ = read_xtc_frames_dev(M.TRAJ, scale=10.0)
    assert np.array_equal(xyz.permute(1, 2, 0).contiguous().cpu().numpy(), coords)
    # contacts through the device-side compaction == threshold on the held distances' own pair order
    s1, s2 = g["distances_sel1"], g["distances_sel2"]
    lists = du.contacts_trajectory(coords, box, s1, s2, g["distances_chains"], False, True, 8.0)
    d = M.run_projection(du, coords, box, g, "distances")
    for f in (0, 57, 199):
        hit = np.nonzero(d[f] <= np.float32(8.0))[0]
        i, j = np.divmod(hit, len(s2))
        # (dist2 <= 64 and sqrt(dist2) <= 8 agree: the root is correctly rounded and 8 is exact)
        assert lists[f] == np.stack([s1[i], s2[j]], 1).ravel().tolist()


# ------------------------------------------------------------------------------------------------
# round 6: k_dist_reduction_closest on the hardware, and the device-resident entry points
# ------------------------------------------------------------------------------------------------
def _ragged(rng, n_atoms, ng, lo, hi):
    return [rng.choice(n_atoms, int(rng.integers(lo, hi + 1)), replace=False).tolist() for _ in range(ng)]


@pytest.mark.parametrize("selfdist,pairs", [(False, False), (True, False), (False, True)])
def test_closest_reduction_many_tiles_every_block_size_bit_exact(selfdist, pairs):
    """Hundreds of (group pair, frame) tiles, ragged groups of 1 ... 21 atoms, 200 frames (a ragged frame tile): 4 / 8 first
    atoms in registers, the generic kernel and the default choice all give the oracle's bits."""
    from moleculekit_amd import _lib
    from moleculekit_amd.distance_utils import dist_trajectory_reduction, dist_trajectory_reduction_pairs
    rng = np.random.default_rng(91 + 2 * selfdist + pairs)
    N, F = 900, 200
    coords = rng.uniform(-40, 40, size=(N, 3, F)).astype(np.float32)
    box = rng.uniform(35, 47, size=(3, F)).astype(np.float32)
    masses = rng.uniform(1, 32, N).astype(np.float32)
    g1 = _ragged(rng, N, 61, 1, 21)
    g2 = g1 if selfdist else _ragged(rng, N, 61 if pairs else 37, 1, 16)
    ch1 = rng.integers(0, 3, len(g1)).astype(np.uint32)
    ch2 = ch1 if selfdist else rng.integers(0, 3, len(g2)).astype(np.uint32)
    ctx = _lib.default_context()
    try:
        for pbc in (True, False):
            want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, selfdist, pbc, masses, 0, 0, pairs=pairs)
            for block in (0, 4, 8, -1, 104, 108):
                ctx.set_reduction_block(block)
                r = np.zeros_like(want)
                if pairs:
                    dist_trajectory_reduction_pairs(coords, box, g1, g2, ch1, ch2, pbc, masses, 0, 0, r)
                else:
                    dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, selfdist, pbc, masses, 0, 0, r)
                assert np.array_equal(r, want), (pbc, block)
    finally:
        ctx.set_reduction_block(0)


def test_closest_reduction_redo_path_and_nan_rules_on_the_hardware():
    """Image integers where rndne(d * fl(1/b)) is not the reference's round(d / b) (the redo path), NaN / inf coordinates and a
    zero box: the same cases as the emulated tier (tests/test_distance_cpu.py), on the real v_rndne / v_min3 / v_max3."""
    from moleculekit_amd import _lib
    from moleculekit_amd.distance_utils import dist_trajectory_reduction
    from tests.test_distance_cpu import _image_integer_traps
    rng = np.random.default_rng(77)
    F, N = 64, 24
    traps = _image_integer_traps(rng, F)
    coords = rng.uniform(0, 12, size=(N, 3, F)).astype(np.float32)
    box = np.empty((3, F), np.float32)
    for f, (b, d) in enumerate(traps):
        ax = f % 3
        box[:, f] = [np.float32(41.3), np.float32(37.9), np.float32(44.1)]
        box[ax, f] = b
        coords[1, :, f] = coords[0, :, f]
        coords[1, ax, f] = np.float32(coords[0, ax, f] - d)
    g1, g2 = [[0], [2, 3, 4, 5, 6]], [[1], [7, 8, 9]]
    ch1, ch2 = np.zeros(2, np.uint32), np.ones(2, np.uint32)
    masses = np.ones(N, np.float32)
    ctx = _lib.default_context()
    try:
        want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, True, masses, 0, 0)
        for block in (4, 8):
            ctx.set_reduction_block(block)
            r = np.zeros_like(want)
            dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, True, masses, 0, 0, r)
            assert np.array_equal(r, want), block
        # NaN / inf / zero box
        coords = rng.uniform(0, 15, size=(30, 3, F)).astype(np.float32)
        box = np.full((3, F), 25.0, np.float32)
        coords[4, 1, ::3] = np.nan; coords[9, 2, ::5] = np.nan; coords[12, 0, ::7] = np.inf
        box[:, 10] = 0.0
        g1 = [[4, 5, 6], [7, 8, 9, 10, 11], [12, 13]]
        g2 = [[14, 15, 16, 17], [18, 9], [20, 21, 22, 23, 24, 25, 26, 27, 28]]
        ch1, ch2 = np.zeros(3, np.uint32), np.ones(3, np.uint32)
        masses = np.ones(30, np.float32)
        with np.errstate(all="ignore"):
            for pbc in (True, False):
                want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, pbc, masses, 0, 0)
                for block in (4, 8, -1):
                    ctx.set_reduction_block(block)
                    r = np.zeros_like(want)
                    dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, pbc, masses, 0, 0, r)
                    assert np.array_equal(r, want, equal_nan=True), (pbc, block)
    finally:
        ctx.set_reduction_block(0)


def test_device_resident_entry_points_bit_exact(g):
    """mkamd_dist_reduction_dev / mkamd_contacts_trajectory_dev / mkamd_cdist_dev / mkamd_pdist_dev: device tensors in, device
    results out on the caller's stream -- the bits of the host forms (and so the reference's)."""
    import torch
    from moleculekit_amd import _lib
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    stream = torch.cuda.Stream(dev)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    c, b, m = g["coords"], g["box"], g["masses"]
    N, F = c.shape[0], c.shape[2]
    g1 = [list(map(int, x)) for x in g["groups1"]]; g2 = [list(map(int, x)) for x in g["groups2"]]
    csr = lambda gs: (np.concatenate([np.asarray(x, np.int32) for x in gs]), np.concatenate([[0], np.cumsum([len(x) for x in gs])]).astype(np.int64))
    a1, o1 = csr(g1); a2, o2 = csr(g2)
    d_c, d_b, d_m = t(c), t(b), t(m)
    d_a1, d_o1, d_a2, d_o2 = t(a1), t(o1), t(a2), t(o2)
    d_ch1, d_ch2 = t(g["gchains1"].astype(np.int32)), t(g["gchains2"].astype(np.int32))
    with torch.cuda.stream(stream):
        ctx.set_stream(stream.cuda_stream)
        try:
            for r1 in (0, 1):
                for r2 in (0, 1):
                    for pbc in (0, 1):
                        want = g[f"red_{r1}{r2}_pbc{pbc}"]
                        out = torch.full(want.shape, -3.0, device=dev, dtype=torch.float32)
                        ctx.dist_reduction_dev(d_c, N, F, d_b, d_a1, d_o1, len(g1), len(a1), d_a2, d_o2, len(g2), d_ch1, d_ch2, False, False, bool(pbc),
                                               d_m, r1, r2, out)
                        stream.synchronize()
                        assert np.array_equal(out.cpu().numpy(), want), (r1, r2, pbc)
            want = g["red_self"]
            out = torch.empty(want.shape, device=dev, dtype=torch.float32)
            ctx.dist_reduction_dev(d_c, N, F, d_b, d_a2, d_o2, len(g2), len(a2), d_a2, d_o2, len(g2), d_ch2, d_ch2, True, False, True, d_m, 0, 0, out)
            stream.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
            # contact lists: the device list == the reference's lists
            s1, s2 = t(g["sel1"].astype(np.int32)), t(g["sel2"].astype(np.int32))
            d_ch = t(g["chains"].astype(np.int32))
            offs, ptr, n = ctx.contacts_trajectory_dev(d_c, F, d_b, s1, len(g["sel1"]), s2, len(g["sel2"]), d_ch, False, True, 12.0)
            assert np.array_equal(np.diff(offs), g["contacts_counts"]) and n == int(g["contacts_counts"].sum()) and ptr
            flat = np.empty(2 * n, np.uint32)
            _lib._check(_lib.load().mkamd_copy_to_host(ctx._h, flat.ctypes.data, ptr, flat.nbytes))
            assert np.array_equal(flat.astype(np.int64), g["contacts_flat"])
            offs, ptr, n = ctx.contacts_trajectory_dev(d_c, F, d_b, s2, len(g["sel2"]), s2, len(g["sel2"]), d_ch, True, False, 15.0)
            flat = np.empty(2 * n, np.uint32)
            _lib._check(_lib.load().mkamd_copy_to_host(ctx._h, flat.ctypes.data, ptr, flat.nbytes))
            assert np.array_equal(np.diff(offs), g["contacts_self_counts"]) and np.array_equal(flat.astype(np.int64), g["contacts_self_flat"])
            only1 = np.setdiff1d(g["sel1"], g["sel2"]).astype(np.int32)          # (an atom in both selections is 0 A from itself)
            offs, ptr, n = ctx.contacts_trajectory_dev(d_c, F, d_b, t(only1), len(only1), s2, len(g["sel2"]), d_ch, False, True, 0.001)
            assert n == 0 and ptr == 0 and not offs.any()
            for D in (1, 2, 3, 5):
                ca, cb = t(g[f"cdist_a{D}"]), t(g[f"cdist_b{D}"])
                out = torch.empty(g[f"cdist_r{D}"].shape, device=dev, dtype=torch.float32)
                ctx.cdist_dev(ca, ca.shape[0], cb, cb.shape[0], D, out)
                po = torch.empty(g[f"pdist_r{D}"].shape, device=dev, dtype=torch.float32)
                ctx.pdist_dev(cb, cb.shape[0], D, po)
                stream.synchronize()
                assert np.array_equal(out.cpu().numpy(), g[f"cdist_r{D}"]) and np.array_equal(po.cpu().numpy(), g[f"pdist_r{D}"])
        finally:
            ctx.set_stream(None)


def test_device_contact_list_equals_the_host_form_on_a_larger_call():
    """37 500 pairs x 300 frames, thousands of contacts: the device-resident list of mkamd_contacts_trajectory_dev read back ==
    the lists of the host form.  (Growth of the device list across several chunks of frames needs a counter budget below the
    library's 256 MB: covered on the emulated tier, tests/test_distance_cpu.py, with a one-byte budget.)"""
    import torch
    from moleculekit_amd import _lib
    from moleculekit_amd.distance_utils import contacts_trajectory
    rng = np.random.default_rng(12)
    N, F = 400, 300
    coords = rng.uniform(0, 30, size=(N, 3, F)).astype(np.float32)
    box = np.full((3, F), 30.0, np.float32)
    chains = (np.arange(N) // 100).astype(np.uint32)
    sel = np.arange(N, dtype=np.uint32)
    want = contacts_trajectory(coords, box, sel[:150], sel[150:], chains, False, True, 4.0)
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_c, d_b = t(coords), t(box)
    offs, ptr, n = ctx.contacts_trajectory_dev(d_c, F, d_b, t(sel[:150].astype(np.int32)), 150, t(sel[150:].astype(np.int32)), 250,
                                               t(chains.astype(np.int32)), False, True, 4.0)
    flat = np.empty(2 * n, np.uint32)
    _lib._check(_lib.load().mkamd_copy_to_host(ctx._h, flat.ctypes.data, ptr, flat.nbytes))
    got = [flat[2 * offs[f]:2 * offs[f + 1]].astype(np.int64).tolist() for f in range(F)]
    assert got == want and n > 1000


@pytest.mark.parametrize("ids", ["selections", "chains", "one"])
def test_rectangular_contact_kernel_on_the_device(ids):
    """k_contacts_count_rect / k_contacts_fill_rect (round 6): the cases of tests/test_distance_cpu.py on the hardware -- rows ending inside a
    run of 16 and inside a tile of 64, ragged first atoms and frames, image-integer traps, a zero box, inf / NaN, every / some / no pair
    wrapping -- against the oracle's squared distances in the reference's (frame, i, j) order; host form and device-resident form."""
    import torch
    from moleculekit_amd import _lib
    from moleculekit_amd.distance_utils import contacts_trajectory
    from tests.test_distance_cpu import _contact_lists, _rect_contact_case
    dev = torch.device("cuda", 0)
    ctx = _lib.default_context(0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    for n2 in (70, 150):
        c, b, ch, s1, s2 = _rect_contact_case(ids, n2)
        F = c.shape[2]
        with np.errstate(all="ignore"):
            for pbc in (True, False):
                d2 = oracle.dist_trajectory(c, b, s1, s2, ch, False, pbc, squared=True)
                for thr in (6.0, 21.5):
                    want = _contact_lists(d2, s1, s2, thr)
                    assert contacts_trajectory(c, b, s1, s2, ch, False, pbc, thr) == want, (n2, pbc, thr)
                    offs, ptr, n = ctx.contacts_trajectory_dev(t(c), F, t(b), t(s1.astype(np.int32)), len(s1), t(s2.astype(np.int32)), len(s2),
                                                               t(ch.astype(np.int32)), False, pbc, thr)
                    flat = np.empty(2 * n, np.uint32)
                    if n:
                        _lib._check(_lib.load().mkamd_copy_to_host(ctx._h, flat.ctypes.data, ptr, flat.nbytes))
                    assert [flat[2 * offs[f]:2 * offs[f + 1]].astype(np.int64).tolist() for f in range(F)] == want


def test_rectangular_contact_kernel_redoes_trapped_batches_on_the_device():
    """A threshold whose square is exactly the reference's d^2 of a trapped pair: a contact only if its batch was redone pair by pair."""
    from moleculekit_amd.distance_utils import contacts_trajectory
    from tests.test_distance_cpu import _trapped_threshold_cases
    cases, ch, s1, s2 = _trapped_threshold_cases()
    for c, b, thr, want in cases:
        assert contacts_trajectory(c, b, s1, s2, ch, False, True, thr) == want


@pytest.mark.parametrize("ids", ["selections", "chains"])
def test_contact_lists_of_few_frames_on_the_device(ids):
    """Calls of at most 16 frames count with lanes along the second atoms (k_contacts_count_rect_few) and keep unpadded mask rows: slices of the
    rectangular case (one frame; the frames with the zero box, inf, NaN; sixteen frames with their traps) against the oracle; get_collisions
    (one frame, no box) at a few thousand atoms against a brute-force count."""
    from moleculekit_amd.distance_utils import contacts_trajectory, get_collisions
    from tests.test_distance_cpu import _contact_lists, _rect_contact_case
    c, b, ch, s1, s2 = _rect_contact_case(ids, 150)
    with np.errstate(all="ignore"):
        for lo, hi in ((0, 1), (10, 14), (0, 16)):
            cs, bs = np.ascontiguousarray(c[:, :, lo:hi]), np.ascontiguousarray(b[:, lo:hi])
            for pbc in (True, False):
                d2 = oracle.dist_trajectory(cs, bs, s1, s2, ch, False, pbc, squared=True)
                for thr in (6.0, 21.5):
                    assert contacts_trajectory(cs, bs, s1, s2, ch, False, pbc, thr) == _contact_lists(d2, s1, s2, thr), (lo, hi, pbc, thr)
    rng = np.random.default_rng(6)
    a, bb = rng.uniform(0, 25, size=(3000, 3)).astype(np.float32), rng.uniform(0, 25, size=(700, 3)).astype(np.float32)
    got = np.asarray(get_collisions(a, bb, 1.5), np.int64).reshape(-1, 2)
    diff = a[:, None, :] - bb[None, :, :]
    exact = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]       # float32, the reference's order (:107-110)
    want = np.argwhere(exact <= np.float32(1.5) * np.float32(1.5))
    assert np.array_equal(got, want) and len(want) > 100


@pytest.mark.parametrize("mixed", [False, True])
def test_pair_table_walk_packed_batches_on_the_device(mixed):
    """for_pair_run's packed batches (round 6) on the hardware: selfdist with an image-integer trap in every frame, a zero box, inf / NaN;
    every atom its own chain, and three chains (mixed batches); distances and the contact lists of the same walk."""
    from moleculekit_amd.distance_utils import contacts_trajectory, dist_trajectory
    from tests.test_distance_cpu import _pair_table_trap_case
    c, b, ch, sel, _ = _pair_table_trap_case(mixed)
    with np.errstate(all="ignore"):
        want = oracle.dist_trajectory(c, b, sel, sel, ch, True, True)
        got = np.zeros_like(want)
        dist_trajectory(c, b, sel, sel, ch, True, True, got)
        assert np.array_equal(got, want, equal_nan=True)
        d2 = oracle.dist_trajectory(c, b, sel, sel, ch, True, True, squared=True)
        res = contacts_trajectory(c, b, sel, sel, ch, True, True, 9.0)
        iu, ju = np.triu_indices(len(sel), 1)
        for f in range(c.shape[2]):
            hit = np.nonzero(d2[f] <= np.float32(81.0))[0]
            assert res[f] == np.stack([sel[iu[hit]], sel[ju[hit]]], 1).astype(np.int64).ravel().tolist(), f


@pytest.mark.parametrize("D", [2, 3, 4])
def test_cdist_pdist_at_size_bit_exact(D):
    """cdist / pdist at sizes that take thousands of blocks of the row kernels (D = 2, 3) and the generic kernels (D = 4): the oracle's
    bits, odd row lengths (16-byte stores at 4-byte alignment), through the host forms."""
    from moleculekit_amd.distance_utils import cdist, pdist
    rng = np.random.default_rng(50 + D)
    a = rng.normal(0, 25, size=(300, D)).astype(np.float32)
    b = rng.normal(0, 25, size=(4099, D)).astype(np.float32)
    r = np.zeros((300, 4099), np.float32)
    cdist(a, b, r)
    assert np.array_equal(r, oracle.cdist(a, b))
    c = rng.normal(0, 25, size=(3001, D)).astype(np.float32)
    r = np.zeros(3001 * 3000 // 2, np.float32)
    pdist(c, r)
    assert np.array_equal(r, oracle.pdist(c))


@pytest.mark.parametrize("mixed", [False, True])
def test_periodic_row_kernel_image_integers_on_the_hardware(mixed):
    """k_dist_rows' periodic rows on the hardware: the emulated tier's trap case (image integers where rndne(d * fl(1/b)) != round(d / b), a
    zero box, inf / NaN coordinates), one chain id among the second atoms and mixed ids, with and without 16-byte stores -- and a bigger
    random call with unwrapped coordinates (quotients up to +-2.5) against the oracle."""
    from moleculekit_amd import _lib
    from moleculekit_amd.distance_utils import dist_trajectory
    from tests.test_distance_cpu import _row_kernel_trap_case
    ctx = _lib.default_context()
    c, b, ch, s1, s2 = _row_kernel_trap_case(mixed)
    try:
        with np.errstate(all="ignore"):
            want = oracle.dist_trajectory(c, b, s1, s2, ch, False, True)
            for avoid in (1, 1 | 8):
                ctx.set_dist_kernels(avoid)
                r = np.zeros_like(want)
                dist_trajectory(c, b, s1, s2, ch, False, True, r)
                assert "k_dist_rows<true" in ctx.last_dist_kernel(), ctx.last_dist_kernel()
                assert np.array_equal(r, want, equal_nan=True), avoid
        ctx.set_dist_kernels(0)
        rng = np.random.default_rng(70 + mixed)
        N, F = 3000, 130
        c = rng.uniform(-40, 40, size=(N, 3, F)).astype(np.float32)             # unwrapped: quotients up to +-2.5
        b = rng.uniform(30, 40, size=(3, F)).astype(np.float32)
        s1 = np.sort(rng.choice(N, 70, replace=False)).astype(np.uint32)
        s2 = np.sort(rng.choice(N, 700, replace=False)).astype(np.uint32)
        ch = (np.arange(N) // 100).astype(np.uint32) if mixed else np.isin(np.arange(N), s2).astype(np.uint32) + 1
        want = oracle.dist_trajectory(c, b, s1, s2, ch, False, True)
        r = np.zeros_like(want)
        dist_trajectory(c, b, s1, s2, ch, False, True, r)
        assert "k_dist_rows<true, 4" in ctx.last_dist_kernel()
        assert np.array_equal(r, want)
    finally:
        ctx.set_dist_kernels(0)


def test_sharded_distances_on_the_device_single_rank():
    """distributed.ShardedDistances on the GPU (one process, no group: the whole trajectory is this rank's shard): the trajectory resident
    in HBM, selections / groups replicated once, every method against the oracle bit for bit; results stay on the device."""
    import torch
    from moleculekit_amd.distributed import ShardedDistances
    rng = np.random.default_rng(23)
    N, F = 500, 90
    coords = rng.uniform(-25, 25, size=(N, 3, F)).astype(np.float32)
    box = rng.uniform(28, 36, size=(3, F)).astype(np.float32)
    chains = (np.arange(N) // 60).astype(np.uint32)
    masses = rng.uniform(1, 32, N).astype(np.float32)
    sd = ShardedDistances.from_host(coords, box)
    assert (sd.lo, sd.hi, sd.n_local) == (0, F, F) and sd.device.type == "cuda"
    s1 = np.sort(rng.choice(N, 40, replace=False)).astype(np.uint32)
    s2 = np.sort(rng.choice(N, 230, replace=False)).astype(np.uint32)
    for selfd, a, b in ((False, s1, s2), (True, s2, s2)):
        for pbc in (True, False):
            got = sd.dist_trajectory(a, b, chains, selfd, pbc)
            assert got.is_cuda and np.array_equal(sd.gather(got).cpu().numpy(), oracle.dist_trajectory(coords, box, a, b, chains, selfd, pbc))
    g1 = [rng.choice(N, int(rng.integers(1, 18)), replace=False).tolist() for _ in range(23)]
    g2 = [rng.choice(N, int(rng.integers(1, 12)), replace=False).tolist() for _ in range(17)]
    ch1, ch2 = rng.integers(0, 3, 23).astype(np.uint32), rng.integers(0, 3, 17).astype(np.uint32)
    for r1, r2 in ((0, 0), (1, 0), (1, 1)):
        got = sd.dist_trajectory_reduction(g1, g2, ch1, ch2, False, True, masses, r1, r2)
        assert np.array_equal(got.cpu().numpy(), oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, False, True, masses, r1, r2)), (r1, r2)
    got = sd.dist_trajectory_reduction(g1, g1, ch1, ch1, True, False, masses, 0, 0)
    assert np.array_equal(got.cpu().numpy(), oracle.dist_trajectory_reduction(coords, box, g1, g1, ch1, ch1, True, False, masses, 0, 0))
    got = sd.dist_trajectory_reduction(g1[:17], g2, ch1[:17], ch2, False, True, masses, 0, 0, pairs=True)
    assert np.array_equal(got.cpu().numpy(), oracle.dist_trajectory_reduction(coords, box, g1[:17], g2, ch1[:17], ch2, False, True, masses, 0, 0, pairs=True))
    offs, pairs = sd.contacts_trajectory(s1, s2, chains, False, True, 9.0)
    d2 = oracle.dist_trajectory(coords, box, s1, s2, chains, False, True, squared=True)
    assert pairs.is_cuda and len(offs) == F + 1
    flat = pairs.cpu().numpy().astype(np.int64)
    for f in (0, 41, F - 1):
        hit = np.nonzero(d2[f] <= np.float32(81.0))[0]
        i, j = np.divmod(hit, len(s2))
        assert np.array_equal(flat[offs[f]:offs[f + 1]], np.stack([s1[i], s2[j]], 1))
    # a second contacts call reuses the context's list: the first result was copied out of it and is still what it was
    keep = flat.copy()
    sd.contacts_trajectory(s2[:50], s2[50:120], chains, False, False, 30.0)
    assert np.array_equal(pairs.cpu().numpy().astype(np.int64), keep)
    with pytest.raises(ValueError):
        sd.dist_trajectory(np.array([N], np.uint32), s2, chains, False, True)


def test_host_calls_that_pack_the_selected_atoms_equal_the_unpacked_calls():
    """csrc/host_pack.h (round 6): a host call whose selections are at most a quarter of a > 1-MB coordinate array uploads only the
    selected atoms' rows, in a packed numbering.  Unsorted selections with repeats, chain ids and masses gathered, contact lists
    translated back to the caller's atom numbers: dist_trajectory, contacts_trajectory and the group reductions (closest atom and
    centre of mass) against the oracle on the whole array, bit for bit."""
    from moleculekit_amd import distance_utils as du
    rng = np.random.default_rng(29)
    N, F = 3000, 40                                              # 1.44 MB: packed
    c = rng.uniform(0, 40.0, size=(N, 3, F)).astype(np.float32)
    b = np.full((3, F), 40.0, np.float32)
    ch = rng.integers(0, 4, size=N).astype(np.uint32)
    m = rng.uniform(1, 32, size=N).astype(np.float32)
    s1 = rng.integers(0, N, size=130).astype(np.uint32); s1[5] = s1[6]          # unsorted, a repeat
    s2 = rng.permutation(N)[:70].astype(np.uint32)
    for selfd, a, bb in ((False, s1, s2), (True, s2, s2)):
        for pbc in (True, False):
            exp = oracle.dist_trajectory(c, b, a, bb, ch, selfd, pbc)
            got = np.full_like(exp, -1.0)
            du.dist_trajectory(c, b, a, bb, ch, selfd, pbc, got)
            assert np.array_equal(got, exp), (selfd, pbc)
        d2 = oracle.dist_trajectory(c, b, a, bb, ch, selfd, True, squared=True)
        table = ([(a[i], bb[j]) for i in range(len(a)) for j in range(i + 1, len(bb))] if selfd else
                 [(a[i], bb[j]) for i in range(len(a)) for j in range(len(bb))])
        lists = du.contacts_trajectory(c, b, a, bb, ch, selfd, True, 9.0)
        n_hits = 0
        for f in range(F):
            hits = np.nonzero(d2[f] <= np.float32(81.0))[0]
            assert lists[f] == [int(v) for k in hits for v in table[k]], (selfd, f)
            n_hits += len(hits)
        assert n_hits > 50
    atoms = rng.permutation(N)[:240]
    g1 = [list(map(int, atoms[i * 8:(i + 1) * 8])) for i in range(12)]
    g2 = [list(map(int, atoms[96 + i * 9:96 + (i + 1) * 9])) for i in range(16)]
    c1 = rng.integers(0, 3, size=12).astype(np.uint32); c2 = rng.integers(0, 3, size=16).astype(np.uint32)
    for r1, r2 in ((0, 0), (1, 1), (0, 1)):
        for pbc in (True, False):
            exp = oracle.dist_trajectory_reduction(c, b, g1, g2, c1, c2, False, pbc, m, r1, r2)
            got = np.full_like(exp, -1.0)
            du.dist_trajectory_reduction(c, b, g1, g2, c1, c2, False, pbc, m, r1, r2, got)
            assert np.array_equal(got, exp), (r1, r2, pbc)


def test_reductions_of_few_frames_take_lanes_along_the_second_groups():
    """k_dist_reduction_few (round 6): calls of at most 8 frames (one structure's residue-contact map) run their lanes along the
    second groups (16 when periodic).  The library's choice at 1 ... 17 frames, against the oracle bit for bit: 300 residue-like groups (several
    blocks along a row; selfdist rows that skip the blocks in front of the diagonal), a first group larger than an LDS pass,
    every reduction mode, periodic with mixed chains and open; the kernels whose lanes are frames beside them; the
    kernel forced onto 70 frames; NaN / zero-box semantics (`dist2 < mindist or mindist < 0`: only the first pair's NaN stays)."""
    from moleculekit_amd import distance_utils as du, _lib
    rng = np.random.default_rng(53)
    N = 4000
    m = rng.uniform(1, 32, size=N).astype(np.float32)
    perm = rng.permutation(N)
    res = [sorted(map(int, perm[i * 11:i * 11 + int(rng.integers(4, 12))])) for i in range(300)]
    chs = (np.arange(300) // 100).astype(np.uint32)
    big = [list(map(int, rng.choice(N, 300, replace=False)))] + res[:3]
    chb = np.array([0, 1, 2, 0], np.uint32)
    for F in (1, 5, 8, 9, 16, 17):
        c = rng.uniform(0, 60.0, size=(N, 3, F)).astype(np.float32)
        b = rng.uniform(40, 60, size=(3, F)).astype(np.float32)
        for pbc in (True, False):
            exp = oracle.dist_trajectory_reduction(c, b, res, res, chs, chs, True, pbc, m, 0, 0)
            got = np.full_like(exp, -1.0)
            du.dist_trajectory_reduction(c, b, res, res, chs, chs, True, pbc, m, 0, 0, got)
            assert np.array_equal(got, exp), (F, pbc)
        for r1, r2 in ((0, 0), (1, 0), (0, 1), (1, 1)):
            exp = oracle.dist_trajectory_reduction(c, b, big, res, chb, chs, False, True, m, r1, r2)
            got = np.full_like(exp, -1.0)
            du.dist_trajectory_reduction(c, b, big, res, chb, chs, False, True, m, r1, r2, got)
            assert np.array_equal(got, exp), (F, r1, r2)
    ctx = _lib.default_context(0)
    F = 70
    c = rng.uniform(0, 60.0, size=(N, 3, F)).astype(np.float32)
    b = rng.uniform(40, 60, size=(3, F)).astype(np.float32)
    b[:, 7] = 0.0
    c[res[0][0], 1, ::3] = np.nan                          # a first atom of a first group: NaN rows
    c[res[5][-1], 2, ::4] = np.nan                         # a later atom: ignored by the minimum
    with np.errstate(all="ignore"):
        exp = oracle.dist_trajectory_reduction(c, b, res[:40], res, chs[:40], chs, False, True, m, 0, 0)
    assert np.isnan(exp).any() and np.isfinite(exp).any()
    try:
        for block in (-2, 0):
            ctx.set_reduction_block(block)
            got = np.full_like(exp, -1.0)
            du.dist_trajectory_reduction(c, b, res[:40], res, chs[:40], chs, False, True, m, 0, 0, got, ctx=ctx)
            assert np.array_equal(got, exp, equal_nan=True), block
    finally:
        ctx.set_reduction_block(0)


def test_dist_trajectory_of_few_frames_takes_lanes_along_atoms():
    """Round 6 (late): on one structure or a handful of frames the kernels whose lanes run along frames are starved.  Rectangular calls of <= 32
    frames take the row kernel wherever it applies; selfdist calls of <= 32 frames with >= 700 atoms (>= 1 500: any frame count) take its triangular form (tasks below the
    diagonal skipped, the reference's condensed order written directly).  Both sides of both thresholds, equal and unequal selections, periodic
    with mixed chains and open: the oracle's bits, the kernel names the library reports, and the pair-table kernel (avoid bit 64) on the same call."""
    from moleculekit_amd import distance_utils as du, _lib
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(71)
    N = 2500
    ch = rng.integers(0, 4, size=N).astype(np.uint32)
    sa = rng.permutation(N)[:900].astype(np.uint32)
    sb = rng.permutation(N)[:720].astype(np.uint32)
    for F in (1, 32, 33):
        c = rng.uniform(0, 50.0, size=(N, 3, F)).astype(np.float32)
        b = rng.uniform(35, 50, size=(3, F)).astype(np.float32)
        for pbc in (True, False):
            for a1, a2 in ((sa, sa), (sb, sa)):
                exp = oracle.dist_trajectory(c, b, a1, a2, ch, True, pbc)
                got = np.full_like(exp, -1.0)
                du.dist_trajectory(c, b, a1, a2, ch, True, pbc, got, ctx=ctx)
                assert np.array_equal(got, exp), ("self", F, pbc, len(a1), len(a2))
                tri = ctx.last_dist_kernel().count(",") == 3                       # k_dist_rows<PBC, JPL, VEC, true>: the triangular form
                assert tri == (F <= 32) and (tri or "k_dist_pairs" in ctx.last_dist_kernel()), (F, ctx.last_dist_kernel())
                if tri:
                    ctx.set_dist_kernels(64)
                    try:
                        du.dist_trajectory(c, b, a1, a2, ch, True, pbc, got, ctx=ctx)
                    finally:
                        ctx.set_dist_kernels(0)
                    assert np.array_equal(got, exp) and "k_dist_pairs" in ctx.last_dist_kernel()
            exp = oracle.dist_trajectory(c, b, sa[:300], sb[:60], ch, False, pbc)           # rows of one atom per lane, a small result
            got = np.full_like(exp, -1.0)
            du.dist_trajectory(c, b, sa[:300], sb[:60], ch, False, pbc, got, ctx=ctx)
            assert np.array_equal(got, exp), ("rect", F, pbc)
            assert ("k_dist_rows" in ctx.last_dist_kernel()) == (F <= 32), (F, ctx.last_dist_kernel())
