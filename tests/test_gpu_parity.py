"""GPU tier (-m gpu): parity tests proper.  Every case goes through the C ABI of libmkamd.so
(ctypes) on the MI355X and is compared with the oracle / the real reference's golden outputs at
|err| <= 1e-5 (BASELINE.json north_star tolerance; float32 GPU path vs float64 reference)."""
import numpy as np
import pytest

from oracle import oracle
from tests.cases import LATTICE_CASES, TOL, check, golden, oracle_lattice
from tests.synth import grid_origin, synth_config

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile_k", [0, 4, 8])
@pytest.mark.parametrize("name", sorted(LATTICE_CASES))
def test_lattice_case(hip_ctx, name, tile_k):
    from moleculekit_amd import batch
    case = LATTICE_CASES[name]()
    hip_ctx.set_tile_k(tile_k)
    try:
        got = batch.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"],
                                     case["nvoxels"], case["voxelsize"], box=case["box"], ctx=hip_ctx)
        hip_ctx.set_force_general(True)
        gen = batch.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"],
                                     case["nvoxels"], case["voxelsize"], box=case["box"], ctx=hip_ctx)
    finally:
        hip_ctx.set_tile_k(0)
        hip_ctx.set_force_general(False)
    check(case, got)
    # class-sorted path (cutoff and w hoisted out of the inner loop) == general per-pair path, bit for bit
    assert np.array_equal(got, gen)


def test_many_sigma_classes_and_dense_tiles(hip_ctx):
    """> 15 distinct sigmas -> general path for the call; > LDS capacity around one tile -> that tile only."""
    from moleculekit_amd import batch
    rng = np.random.default_rng(31)
    n = 400
    c = rng.normal(0, 4, size=(n, 3)).astype(np.float32)
    s = np.where(rng.random((n, 8)) < 0.3, rng.uniform(0.8, 2.5, size=(n, 8)), 0.0)
    o, nv = np.array([[-8.0, -8, -8]]), np.array([16, 16, 16])
    got = batch.voxelize_lattice(c, [0, n], s, o, nv, 1.0, ctx=hip_ctx)
    assert np.abs(got - oracle_lattice(c, np.array([0, n]), s, o, nv, 1.0)).max() <= TOL
    n = 900
    c = rng.uniform(0, 7, size=(n, 3)).astype(np.float32)
    s = np.tile(rng.choice([1.1, 1.7, 1.52], size=(n, 1)), (1, 8))
    o, nv = np.array([[-4.0, -4, -4]]), np.array([24, 16, 16])
    got = batch.voxelize_lattice(c, [0, n], s, o, nv, 1.0, ctx=hip_ctx)
    assert np.abs(got - oracle_lattice(c, np.array([0, n]), s, o, nv, 1.0)).max() <= TOL


@pytest.mark.parametrize("n,chan_prob", [
    (250, [1.0] * 8),                                   # several class-sorted rounds per dense tile
    (110, [1.0] * 8),                                   # over the 640 and 768 tiers, under the 1024 one
    (900, [1.0, 0.1, 0.1, 0.3, 1.0, 0.05, 0.0, 0.2]),   # two channels too dense even on their own
])
def test_dense_tiles_all_lds_tiers_bit_identical(hip_ctx, n, chan_prob):
    """Tiles with more entries than the LDS arrays hold go through the multi-round dense kernel; whatever
    the tier (forced or adaptive) the values are the general path's, bit for bit, and within TOL of the
    oracle (tests/test_emu_kernels.py holds the same cases for the CPU tier)."""
    from moleculekit_amd import batch
    rng = np.random.default_rng(32)
    c = rng.uniform(0, 7, size=(n, 3)).astype(np.float32)
    s = np.tile(rng.choice([1.1, 1.7, 1.52], size=(n, 1)), (1, 8))
    s = np.where(rng.random((n, 8)) < np.asarray(chan_prob)[None, :], s, 0.0)
    o, nv = np.array([[-4.0, -4, -4]]), np.array([24, 16, 16])
    want = oracle_lattice(c, np.array([0, n]), s, o, nv, 1.0)
    outs = []
    try:
        for tier in (0, 1, 2, -1, -1):
            hip_ctx.set_lds_tier(tier)
            outs.append(batch.voxelize_lattice(c, [0, n], s, o, nv, 1.0, ctx=hip_ctx))
        hip_ctx.set_force_general(True)
        outs.append(batch.voxelize_lattice(c, [0, n], s, o, nv, 1.0, ctx=hip_ctx))
    finally:
        hip_ctx.set_lds_tier(-1)
        hip_ctx.set_force_general(False)
    assert np.abs(outs[0] - want).max() <= TOL
    for other in outs[1:]:
        assert np.array_equal(outs[0], other)


@pytest.mark.parametrize("C", [1, 3, 11])
def test_explicit_centres_golden(hip_ctx, C):
    from moleculekit_amd import batch
    g = golden(f"explicit_C{C}.npz")
    got = batch.occupancy_centers(g["centers"], g["coords"], g["sigmas"], ctx=hip_ctx)
    assert np.abs(got - g["features"]).max() <= TOL


def test_explicit_centres_pbc(hip_ctx):
    from moleculekit_amd import batch
    rng = np.random.default_rng(3)
    c = rng.uniform(-20, 40, size=(700, 3)).astype(np.float32)
    s = rng.choice([0, 1.1, 1.7, 1.8], size=(700, 8)).astype(np.float32)
    centers = rng.uniform(0, 15, size=(1000, 3))
    box = np.array([15.0, 17.0, 13.5])
    got = batch.occupancy_centers(centers, c, s, box=box, ctx=hip_ctx)
    exp = oracle.calculate_occupancy(centers, c, s.astype(np.float64), box=box)
    assert np.abs(got - exp).max() <= TOL


def test_calculate_occupancy_dropin_in_place_max(hip_ctx):
    """Exact call contract of occupancy_utils.pyx:34-61, incl. max-accumulation into `results`."""
    from moleculekit_amd.occupancy_utils import calculate_occupancy
    g = golden("explicit_C3.npz")
    centers = np.ascontiguousarray(g["centers"]); coords = np.ascontiguousarray(g["coords"], np.float32)
    sig = np.ascontiguousarray(g["sigmas"])
    res = np.zeros((centers.shape[0], 3))
    assert calculate_occupancy(centers, coords, sig, res, ctx=hip_ctx) is None
    assert np.abs(res - g["features"]).max() <= TOL
    pre = np.full_like(res, 0.5)
    calculate_occupancy(centers, coords, sig, pre, ctx=hip_ctx)
    assert np.abs(pre - np.maximum(0.5, g["features"])).max() <= TOL
    # PBC composition trick of SURVEY 8c works on the drop-in too: accumulate shifted calls
    acc = np.zeros_like(res)
    calculate_occupancy(centers, coords, sig, acc, ctx=hip_ctx)
    calculate_occupancy(centers + 3.0, coords, sig, acc, ctx=hip_ctx)
    ref = np.zeros_like(res)
    oracle.calculate_occupancy(centers, coords, sig, ref)
    oracle.calculate_occupancy(centers + 3.0, coords, sig, ref)
    assert np.abs(acc - ref).max() <= TOL


def test_calculate_occupancy_recognises_the_lattice_its_caller_passes(hip_ctx):
    """The literal C-level call (what the stub of INTEGRATION.md binds): a getCenters lattice -- the reference's only
    usage, voxeldescriptors.py:356 -- must reach the tile kernels (seen through the context's tile-kernel launch
    counter), the same centres jittered must not, and both must agree with the oracle."""
    from moleculekit_amd.occupancy_utils import calculate_occupancy
    from moleculekit_amd.voxeldescriptors import getCenters
    g = golden("cfg1_3ptb.npz")
    coords = np.ascontiguousarray(g["coords"], np.float32); sig = np.ascontiguousarray(g["sigmas"], np.float64)
    centers, _ = getCenters(boxsize=[24, 24, 24], center=np.asarray(g["center"]), voxelsize=1)
    jitter = centers + np.random.default_rng(0).normal(0, 1e-3, centers.shape)
    hip_ctx.enable_kernel_timing(True)
    try:
        hip_ctx.read_kernel_timing()
        res = np.zeros((centers.shape[0], sig.shape[1]))
        calculate_occupancy(centers, coords, sig, res, ctx=hip_ctx)
        assert hip_ctx.read_kernel_timing()[1] >= 1                      # tile-kernel launches
        assert np.abs(res - g["features"]).max() <= TOL
        arb = np.zeros_like(res)
        calculate_occupancy(jitter, coords, sig, arb, ctx=hip_ctx)
        assert hip_ctx.read_kernel_timing()[1] == 0
    finally:
        hip_ctx.enable_kernel_timing(False)
    sub = np.random.default_rng(1).choice(centers.shape[0], 1500, replace=False)
    want = np.zeros((len(sub), sig.shape[1]))
    oracle.calculate_occupancy(np.ascontiguousarray(jitter[sub]), coords, sig, want)
    assert np.abs(arb[sub] - want).max() <= TOL


def test_grid_centers_bit_exact(hip_ctx):
    from moleculekit_amd import batch
    g = golden("getcenters_cases.npz")
    for i in range(5):
        bb = g[f"box{i}_center"] - g[f"box{i}_boxsize"] / 2
        got = batch.grid_centers(bb, g[f"box{i}_nvoxels"], float(g[f"box{i}_voxelsize"]), ctx=hip_ctx)
        assert np.array_equal(got, g[f"box{i}_centers"])


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "pbc_batch", "channels11", "cutoff_adversarial_1A", "voxel15", "special_sigmas"])
def test_team_of_waves_per_tile_is_bit_identical(hip_ctx, name):
    """One grid per call runs four waves per tile (mkamd_ctx_set_tile_team): not a bit may differ from one wave per tile."""
    from moleculekit_amd import batch
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    try:
        for k in (8, 4):
            hip_ctx.set_tile_k(k)
            hip_ctx.set_tile_team(0)
            one = batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx)
            hip_ctx.set_tile_team(1)
            team = batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx)
            assert np.array_equal(one, team)
            check(case, team)
            hip_ctx.set_force_general(True)
            gen = batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx)
            hip_ctx.set_force_general(False)
            assert np.array_equal(gen, one)
    finally:
        hip_ctx.set_tile_k(0); hip_ctx.set_tile_team(-1); hip_ctx.set_force_general(False)


def test_cfg2_full_size_against_reference_samples(hip_ctx):
    """BASELINE.json configs[1] at FULL size (50k atoms, 64^3 x 8): 16384 sampled voxels + per-channel
    checksums of the real reference's output (tests/golden/cfg2_sampled.npz)."""
    from moleculekit_amd import batch
    g = golden("cfg2_sampled.npz")
    p = synth_config(2, 1)
    o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    for k in (4, 8):
        hip_ctx.set_tile_k(k)
        got = batch.voxelize_lattice(p["coords"], p["atom_offsets"], p["sigmas"], o[None], nv, p["voxelsize"],
                                     ctx=hip_ctx)[0].astype(np.float64)
        hip_ctx.set_tile_k(0)
        assert np.abs(got[g["sample_idx"]] - g["sample_features"]).max() <= TOL
        assert np.all(np.abs(got.sum(0) - g["channel_sums"]) <= 1e-6 * got.shape[0])   # mean |err| << 1e-6
        assert np.array_equal(np.count_nonzero(got > 1e-6, axis=0) > 0, g["nonzero"] > 0)
        assert np.abs(got.max(0) - g["channel_max"]).max() <= TOL


def test_cfg4_full_size_against_reference_samples(hip_ctx):
    """BASELINE.json configs[3] at its REAL shape (30 000 atoms per frame, 66.9 A periodic box, 48^3 x 8): two frames in
    one call; 16384 sampled voxels per frame + per-channel checksums of the real reference's 27-image composition
    (tests/golden/cfg4_full_sampled.npz, made by tests/golden/make_golden_cfg4.py)."""
    from moleculekit_amd import batch
    g = golden("cfg4_full_sampled.npz")
    p = synth_config(4, 2)
    origins = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    assert np.array_equal(nv, g["nvoxels"])
    for k in (8, 4):
        hip_ctx.set_tile_k(k)
        got = batch.voxelize_lattice(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, p["voxelsize"], box=p["box"],
                                     ctx=hip_ctx).astype(np.float64)
        hip_ctx.set_tile_k(0)
        for frame in (0, 1):
            f = got[frame]
            assert np.abs(f[g[f"f{frame}_sample_idx"]] - g[f"f{frame}_sample_features"]).max() <= TOL
            assert np.all(np.abs(f.sum(0) - g[f"f{frame}_channel_sums"]) <= 1e-6 * f.shape[0])   # mean |err| << 1e-6
            assert np.array_equal(np.count_nonzero(f > 1e-6, axis=0) > 0, g[f"f{frame}_nonzero"] > 0)
            assert np.abs(f.max(0) - g[f"f{frame}_channel_max"]).max() <= TOL


def test_cfg4_full_count_in_one_call(hip_ctx):
    """BASELINE configs[3] at its FULL count on one GPU: 10 000 periodic frames of 30 000 atoms (3e8 atoms, a 66.9 A box,
    48^3 x 8: 8.8e9 result elements, 35 GB) in ONE call, generated and kept on the device.  Frames at both ends and around
    the 2^32 / 2^33-element marks must carry the bits the same frames get as a small batch (whose path the golden samples
    of the test above pin), and no frame may be left unwritten (the buffer is pre-filled with NaN)."""
    import torch
    from moleculekit_amd import batch
    from tests.synth import synth_sigmas
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(0).total_memory < 128 * 2 ** 30:
        pytest.skip("needs 128 GB of device memory")
    F, N, L = 10_000, 30_000, 66.9
    p1 = synth_config(4, 1)
    origin, nv = grid_origin(p1["centers"][0], p1["boxsize"], p1["voxelsize"])
    V = int(np.prod(nv))
    gen = torch.Generator(device=dev).manual_seed(4)
    x0 = torch.rand((N, 3), generator=gen, device=dev, dtype=torch.float32) * L
    walk = torch.randn((F, N, 3), generator=gen, device=dev, dtype=torch.float32) * 0.3
    walk[0] = 0
    coords = torch.remainder(x0 + torch.cumsum(walk, 0), L).reshape(F * N, 3).contiguous()
    del walk
    sig1 = synth_sigmas(np.random.default_rng(4), N).astype(np.float32)
    sigmas = torch.as_tensor(sig1, device=dev).repeat(F, 1)
    offsets = torch.arange(F + 1, device=dev, dtype=torch.int64) * N
    origins = torch.as_tensor(np.tile(origin, (F, 1)), device=dev, dtype=torch.float64)
    box = torch.full((F, 3), L, device=dev, dtype=torch.float32)
    mi = batch.max_images_per_atom(np.full((1, 3), L), nv, p1["voxelsize"])
    out = torch.full((F, V, 8), float("nan"), dtype=torch.float32, device=dev)
    hip_ctx.set_tile_k(8)                                        # (both calls at one tile depth: the same roundings)
    try:
        batch.voxelize_lattice_torch(coords, offsets, sigmas, origins, nv, p1["voxelsize"], box=box, max_images=mi, ctx=hip_ctx, out=out)
        hip_ctx.synchronize()
        torch.cuda.synchronize()
        sums = out.sum(dim=(1, 2))
        assert bool(torch.isfinite(sums).all()) and float(sums.min()) > 0.0
        per_frame = V * 8
        pick = sorted({0, 1, F // 2, F - 2, F - 1, 2 ** 32 // per_frame, 2 ** 32 // per_frame + 1, 2 ** 33 // per_frame, 2 ** 33 // per_frame + 1})
        idx = torch.as_tensor(pick, device=dev)
        sub_coords = coords.reshape(F, N, 3)[idx].reshape(-1, 3).cpu().numpy()
        small = batch.voxelize_lattice(sub_coords, np.arange(len(pick) + 1) * N, np.tile(sig1, (len(pick), 1)), np.tile(origin, (len(pick), 1)),
                                       nv, p1["voxelsize"], box=np.full((len(pick), 3), L, dtype=np.float32), ctx=hip_ctx)
        big = out[idx].cpu().numpy()
        assert np.array_equal(small, big), float(np.abs(small - big).max())
        assert small.max() > 0.5
    finally:
        hip_ctx.set_tile_k(0)
    del out, coords, sigmas
    torch.cuda.empty_cache()


def test_one_grid_past_2_to_the_32_elements(hip_ctx):
    """ONE item whose result passes 2^32 elements: an 832^3 x 8 grid (4.6e9 floats, 18 GB), 11.5 M atoms, generated on the
    device.  64^3 windows at the first and the last corner and in the middle must carry the bits the same window gets as a
    grid of its own (origin moved by whole cells, the atoms within reach): voxel indices are 64-bit where they have to be."""
    import torch
    from moleculekit_amd import batch
    from tests.synth import synth_sigmas
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(0).total_memory < 64 * 2 ** 30:
        pytest.skip("needs 64 GB of device memory")
    n, N = 832, 11_500_000
    assert n ** 3 * 8 > 2 ** 32
    gen = torch.Generator(device=dev).manual_seed(8)
    coords = (torch.rand((N, 3), generator=gen, device=dev, dtype=torch.float32) * (n + 10.0) - 5.0).contiguous()
    sig1 = synth_sigmas(np.random.default_rng(8), 1_000_000).astype(np.float32)
    sigmas = torch.as_tensor(sig1, device=dev).repeat(12, 1)[:N].contiguous()
    nv = np.array([n, n, n])
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    out = torch.full((1, n ** 3, 8), float("nan"), dtype=torch.float32, device=dev)
    hip_ctx.set_tile_k(8)
    try:
        batch.voxelize_lattice_torch(coords, t([0, N], np.int64), sigmas, t(np.zeros((1, 3)), np.float64), nv, 1.0, ctx=hip_ctx, out=out)
        hip_ctx.synchronize()
        torch.cuda.synchronize()
        grid = out.view(n, n, n, 8)
        assert bool(torch.isfinite(grid.sum(dim=(1, 2, 3))).all())
        for lo in (0, 384, n - 64):
            near = ((coords > lo - 6.0) & (coords < lo + 64 + 6.0)).all(dim=1)
            c, sg = coords[near].cpu().numpy(), sigmas[near].cpu().numpy()
            win = batch.voxelize_lattice(c, np.array([0, len(c)]), sg, np.full((1, 3), float(lo)), np.array([64, 64, 64]), 1.0, ctx=hip_ctx)
            big = grid[lo:lo + 64, lo:lo + 64, lo:lo + 64].reshape(1, 64 ** 3, 8).cpu().numpy()
            assert np.array_equal(win, big), (lo, float(np.abs(win - big).max()))
            assert win.max() > 0.5
    finally:
        hip_ctx.set_tile_k(0)
    del out, grid, coords, sigmas
    torch.cuda.empty_cache()


def test_3ptb_bbox_buffer8_against_reference_samples(hip_ctx):
    """The reference test's own call shape (test_voxeldescriptors.py:77-79: buffer=8 -> 60x55x65 grid)."""
    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    from tests.test_host_logic import Mol
    g = golden("3ptb_bbox_buffer8.npz")
    mol = Mol(g["coords"], element=g["element"])
    feats, centers, nvox = getVoxelDescriptors(mol, buffer=8, voxelsize=1, userchannels=g["userchannels"])
    assert feats.dtype == np.float64 and feats.shape == (int(np.prod(nvox)), 8)
    assert np.array_equal(nvox, g["nvoxels"])
    assert np.array_equal(centers[:4], g["centers_first"]) and np.array_equal(centers[-4:], g["centers_last"])
    assert np.abs(feats[g["sample_idx"]] - g["sample_features"]).max() <= TOL
    assert np.all(np.abs(feats.sum(0) - g["channel_sums"]) <= 1e-6 * feats.shape[0])


def test_full_size_batches_size_independent_properties(hip_ctx):
    """cfg3 at full batch (1024 poses): oracle on a handful of items + properties that hold at any size:
    permutation equivariance over items, independence of batch composition, translation covariance,
    range [0,1], and max-linearity (voxelizing A u B == max(vox A, vox B))."""
    from moleculekit_amd import batch
    B = 1024
    p = synth_config(3, B)
    origins = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    full = batch.voxelize_lattice(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, p["voxelsize"], ctx=hip_ctx)
    assert full.shape == (B, 24 ** 3, 8) and np.all(np.isfinite(full)) and full.min() >= 0 and full.max() <= 1
    # the oracle on 96 of the 1024 items: both ends + 92 seeded random picks (a 24^3 ligand grid costs it ~10 ms)
    pick = sorted({0, 1, 511, 1023} | set(np.random.default_rng(33).choice(B, 92, replace=False).tolist()))
    assert len(pick) >= 64
    for b in pick:
        s, e = p["atom_offsets"][b], p["atom_offsets"][b + 1]
        exp = oracle_lattice(p["coords"][s:e], np.array([0, e - s]), p["sigmas"][s:e], origins[b:b + 1], nv, p["voxelsize"])
        assert np.abs(full[b] - exp[0]).max() <= TOL
    # batch composition / order must not matter: reversed batch gives reversed output, bit for bit
    n = 60
    rev_coords = p["coords"].reshape(B, n, 3)[::-1].reshape(-1, 3)
    rev_sig = p["sigmas"].reshape(B, n, 8)[::-1].reshape(-1, 8)
    rev = batch.voxelize_lattice(rev_coords, p["atom_offsets"], rev_sig, origins[::-1], nv, p["voxelsize"], ctx=hip_ctx)
    assert np.array_equal(rev[::-1], full)
    # translation covariance: with coordinates on a 1/1024 A lattice, shifting atoms and origin by whole
    # voxels is exact in float32, so the result must not change at all (positions are handled relative
    # to the grid, never as absolute float32 numbers)
    shift = np.array([16.0, -8.0, 32.0])
    snapped = (np.round(p["coords"][: 8 * n] * 1024) / 1024).astype(np.float32)
    base = batch.voxelize_lattice(snapped, p["atom_offsets"][:9], p["sigmas"][: 8 * n], origins[:8], nv,
                                  p["voxelsize"], ctx=hip_ctx)
    moved = batch.voxelize_lattice((snapped + shift).astype(np.float32), p["atom_offsets"][:9],
                                   p["sigmas"][: 8 * n], origins[:8] + shift, nv, p["voxelsize"], ctx=hip_ctx)
    assert np.array_equal((snapped + shift).astype(np.float32) - shift.astype(np.float32), snapped)
    assert np.array_equal(moved, base)
    # max-linearity: A u B == max(A, B), exactly (min/max are exact; same tile depth -> same roundings)
    hip_ctx.set_tile_k(8)
    a = batch.voxelize_lattice(p["coords"][:30], np.array([0, 30]), p["sigmas"][:30], origins[:1], nv, 1.0, ctx=hip_ctx)
    b = batch.voxelize_lattice(p["coords"][30:60], np.array([0, 30]), p["sigmas"][30:60], origins[:1], nv, 1.0, ctx=hip_ctx)
    hip_ctx.set_tile_k(0)
    assert np.array_equal(np.maximum(a, b)[0], full[0])


def test_cfg5_full_resolution_batch(hip_ctx):
    """cfg5 shape (0.5 A voxels, ragged molecules) on a 2048-molecule batch, oracle on 96 of them (ends + seeded picks)."""
    from moleculekit_amd import batch
    B = 2048
    p = synth_config(5, B)
    origins = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    full = batch.voxelize_lattice(p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, p["voxelsize"], ctx=hip_ctx)
    pick = sorted({0, 777, 2047} | set(np.random.default_rng(55).choice(B, 93, replace=False).tolist()))
    assert len(pick) >= 64
    for b in pick:
        s, e = p["atom_offsets"][b], p["atom_offsets"][b + 1]
        exp = oracle_lattice(p["coords"][s:e], np.array([0, e - s]), p["sigmas"][s:e], origins[b:b + 1], nv, p["voxelsize"])
        assert np.abs(full[b] - exp[0]).max() <= TOL


def test_cfg5_full_count_in_one_call(hip_ctx):
    """BASELINE configs[4] at its FULL count on one GPU: 100 000 ragged molecules, 24^3 grid at 0.5 A, one call -- 1.1e10
    result elements (44 GB, past 2^33: every index of the path has to be 64-bit where it counts), result kept on the
    device.  The oracle on items at both ends and around the 2^32 / 2^33-element marks, bitwise agreement with the same
    molecules voxelized as a small batch, and nothing left unwritten (the buffer is pre-filled with NaN)."""
    import torch
    from moleculekit_amd import batch
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(0).total_memory < 64 * 2 ** 30:
        pytest.skip("needs 64 GB of device memory")
    B = 100_000
    p = synth_config(5, B)
    origins = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    V = int(np.prod(nv))
    assert B * V * 8 > 2 ** 33
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    out = torch.full((B, V, 8), float("nan"), dtype=torch.float32, device=dev)
    hip_ctx.set_tile_k(8)                                        # (both calls at one tile depth: the same roundings)
    res = batch.voxelize_lattice_torch(t(p["coords"], np.float32), t(p["atom_offsets"], np.int64), t(p["sigmas"], np.float64),
                                       t(origins, np.float64), nv, p["voxelsize"], ctx=hip_ctx, out=out)   # (float64 sigmas, as below)
    hip_ctx.synchronize()
    torch.cuda.synchronize()
    assert res.data_ptr() == out.data_ptr()
    sums = out.sum(dim=(1, 2))                                   # one number per molecule: NaN anywhere would show
    assert bool(torch.isfinite(sums).all()) and float(sums.min()) > 0.0
    per_item = V * 8
    marks = [2 ** 32 // per_item, 2 ** 32 // per_item + 1, 2 ** 33 // per_item, 2 ** 33 // per_item + 1]
    pick = sorted({0, 1, B // 2, B - 2, B - 1, *marks} | set(np.random.default_rng(5).choice(B, 8, replace=False).tolist()))
    for b in pick:
        s, e = p["atom_offsets"][b], p["atom_offsets"][b + 1]
        exp = oracle_lattice(p["coords"][s:e], np.array([0, e - s]), p["sigmas"][s:e], origins[b:b + 1], nv, p["voxelsize"])
        got = out[b].cpu().numpy()
        assert np.abs(got - exp[0]).max() <= TOL, b
    # the last 64 molecules again as a batch of their own: the same bits
    s0 = p["atom_offsets"][B - 64]
    small = batch.voxelize_lattice(p["coords"][s0:], p["atom_offsets"][B - 64:] - s0, p["sigmas"][s0:], origins[B - 64:], nv,
                                   p["voxelsize"], ctx=hip_ctx)
    hip_ctx.set_tile_k(0)
    tail = out[B - 64:].cpu().numpy()
    assert np.array_equal(small, tail), float(np.abs(small - tail).max())
    del out, res
    torch.cuda.empty_cache()


def test_periodic_frames_match_27_image_composition_on_gpu(hip_ctx):
    """PBC result == max over the 27 shifted NON-periodic GPU runs (the composition SURVEY 8c uses to
    pin the extension on the reference kernel), when the grid lies inside the box."""
    from moleculekit_amd import batch
    rng = np.random.default_rng(9)
    L = np.array([30.0, 28.0, 33.0], np.float32)
    n = 2500
    from tests.synth import synth_sigmas
    c = (rng.uniform(0, 1, size=(n, 3)) * L).astype(np.float32)
    s = synth_sigmas(rng, n)
    origin = np.array([[2.0, 1.0, 3.0]])
    nv = [24, 24, 24]
    per = batch.voxelize_lattice(c, [0, n], s, origin, nv, 1.0, box=L[None], ctx=hip_ctx)[0]
    acc = np.zeros_like(per)
    for kx in (-1, 0, 1):
        for ky in (-1, 0, 1):
            for kz in (-1, 0, 1):
                sh = (c.astype(np.float64) + np.array([kx, ky, kz]) * L.astype(np.float64)).astype(np.float32)
                acc = np.maximum(acc, batch.voxelize_lattice(sh, [0, n], s, origin, nv, 1.0, ctx=hip_ctx)[0])
    assert np.abs(per - acc).max() <= 2e-6     # float32 rounding of the shifted coordinates only


def test_empty_and_degenerate_inputs(hip_ctx):
    from moleculekit_amd import batch
    z = batch.voxelize_lattice(np.zeros((0, 3), np.float32), [0], np.zeros((0, 8)), np.zeros((0, 3)), [8, 8, 8], 1.0, ctx=hip_ctx)
    assert z.shape == (0, 512, 8)
    z = batch.voxelize_lattice(np.zeros((0, 3), np.float32), [0, 0], np.zeros((0, 8)), [[0, 0, 0]], [5, 6, 7], 1.0, ctx=hip_ctx)
    assert z.shape == (1, 210, 8) and not z.any()                   # no atoms -> all-zero grid, fully written
    z = batch.voxelize_lattice(np.zeros((1, 3), np.float32), [0, 1], np.ones((1, 8)), [[0, 0, 0]], [4, 0, 4], 1.0, ctx=hip_ctx)
    assert z.shape == (1, 0, 8)
    with pytest.raises(ValueError):
        batch.voxelize_lattice(np.zeros((1, 3), np.float32), [0, 1], np.ones((1, 8)), [[0, 0, 0]], [4, 4, 4], -1.0, ctx=hip_ctx)
    from moleculekit_amd._lib import MkamdError
    with pytest.raises(MkamdError):
        batch.voxelize_lattice(np.zeros((1, 3), np.float32), [0, 1], np.ones((1, 8)), [[0, 0, 0]], [4, 4, 4], 1.0,
                               box=np.array([[9.0, 30, 30]], np.float32), ctx=hip_ctx)


def test_pipelined_calls_match_in_order_calls(hip_ctx):
    """Opt-in cross-call pipelining (pre-pass of call n+1 beside the tile kernel of call n, double-buffered
    workspace) must not change a single bit, also when call sizes alternate and small in-order calls are mixed in."""
    import torch
    from moleculekit_amd import batch
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    work = []
    for i, B in enumerate((6, 1, 5, 6, 4)):
        p = synth_config(2, B, seed=50 + i)
        o = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
        nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
        work.append((t(p["coords"], np.float32), t(p["atom_offsets"], np.int64), t(p["sigmas"], np.float32),
                     t(o, np.float64), nv, p["voxelsize"]))
    def run(pipelined):
        hip_ctx.set_pipelining(pipelined)
        outs = [batch.voxelize_lattice_torch(*w, ctx=hip_ctx) for w in work for _ in range(2)]
        torch.cuda.synchronize()
        hip_ctx.synchronize()
        hip_ctx.set_pipelining(False)
        return [o.cpu().numpy() for o in outs]
    ref = run(False)
    got = run(True)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    assert ref[0].max() > 0.5


def test_many_dense_tiles_with_fixups_in_one_tail_launch(hip_ctx):
    """k_tail at scale: 400 copies of the dense + wide-sigma item (4 800 tiles, most of them dense: more than the 4 096
    dense blocks of the launch, so the blocks loop) -- every fix-up wave has to wait for the dense blocks, and every
    copy must come out with the same bits, within tolerance of the oracle.  In order and with the
    steps software-pipelined (the next call's pre-pass beside this call's tail)."""
    from moleculekit_amd import batch
    from tests.cases import case_dense_with_wide_sigmas, check
    case = case_dense_with_wide_sigmas()
    n = len(case["coords"])
    B = 400
    one = batch.voxelize_lattice(case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"],
                                 case["voxelsize"], ctx=hip_ctx)
    check(case, one)
    coords = np.tile(case["coords"], (B, 1))
    sig = np.tile(case["sigmas"], (B, 1))
    offs = np.arange(B + 1, dtype=np.int64) * n
    origins = np.tile(case["origins"], (B, 1))
    try:
        for pipelined, prepass in ((False, -1), (True, -1), (False, 0), (True, 0)):   # per-item pre-pass / kernel chain
            hip_ctx.set_pipelining(pipelined)
            hip_ctx.set_prepass_mode(prepass)
            for _ in range(3):
                got = batch.voxelize_lattice(coords, offs, sig, origins, case["nvoxels"], case["voxelsize"], ctx=hip_ctx)
                assert got.shape[0] == B
                check(case, got[:1])
                assert np.abs(got[0] - one[0]).max() <= 6e-6      # (the one-item call runs K = 4 tiles, the batch K = 8)
                bad = [b for b in range(B) if not np.array_equal(got[b], got[0])]
                assert not bad, (pipelined, prepass, len(bad), bad[:5], float(np.abs(got[bad[0]] - got[0]).max()))
    finally:
        hip_ctx.set_pipelining(False)
        hip_ctx.set_prepass_mode(-1)


def test_batches_of_ligand_sized_items_take_the_workgroup_per_item_kernel(hip_ctx):
    """2 000 items of 10 .. 70 atoms (the automatic choice: <= 96 atoms on average, per-item pre-pass):
    k_voxelize_items against the wave-per-tile kernel, bit for bit, a sample of the items against the oracle; one item
    of 900 atoms in the middle of the batch (more than 384 entries: the unsorted walk) and an empty one."""
    from moleculekit_amd import batch
    rng = np.random.default_rng(91)
    ns = rng.integers(10, 71, size=2000)
    ns[777] = 900
    ns[778] = 0
    coords = [rng.normal(0, 3.0 if n < 100 else 6.0, size=(n, 3)).astype(np.float32) for n in ns]
    sig = [np.where(rng.random((n, 8)) < [0.5, 0.1, 0.2, 0.1, 0.05, 0.05, 0.01, 0.7], rng.choice([1.2, 1.52, 1.7, 1.9], size=(n, 1)), 0.0) for n in ns]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    origins = np.tile(np.array([[-10.0, -10.0, -10.0]]), (len(ns), 1)) + rng.uniform(-0.5, 0.5, size=(len(ns), 3))
    nv = np.array([20, 20, 20])
    args = (np.concatenate(coords), offs, np.concatenate(sig), origins, nv, 1.0)
    try:
        hip_ctx.set_tile_items(0)
        ref = batch.voxelize_lattice(*args, ctx=hip_ctx)
        hip_ctx.set_tile_items(-1)
        got = batch.voxelize_lattice(*args, ctx=hip_ctx)
    finally:
        hip_ctx.set_tile_items(-1)
    assert np.array_equal(got, ref)
    assert np.all(got[778] == 0.0)
    for b in (0, 1, 777, 1999):
        s_, e_ = offs[b], offs[b + 1]
        want = oracle_lattice(args[0][s_:e_], np.array([0, e_ - s_]), args[2][s_:e_], origins[b:b + 1], nv, 1.0)
        assert np.abs(got[b] - want[0]).max() <= TOL


@pytest.mark.parametrize("name", sorted(LATTICE_CASES))
def test_tolerance_aware_reach_is_opt_in_and_bounded(hip_ctx, name):
    """mkamd_ctx_set_value_tolerance: off by default; with eps in (0, 1e-5] atoms are culled per tile
    where they are worth less than eps -- no value moves by more than eps against the exact mode (and only downwards),
    every case stays inside the 1e-5 parity bound, and switching it off again restores the exact bits."""
    from moleculekit_amd import batch
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    exact = batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx)
    try:
        for eps in (1e-6, 5e-6):
            hip_ctx.set_value_tolerance(eps)
            tol = batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx)
            assert np.abs(tol - exact).max() <= eps * 1.001 and np.all(tol <= exact)
            check(case, tol)
        with pytest.raises(Exception):
            hip_ctx.set_value_tolerance(1e-3)                  # beyond the parity bound: refused
    finally:
        hip_ctx.set_value_tolerance(0.0)
    assert np.array_equal(batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx), exact)


def test_tolerance_aware_reach_on_full_size_cfg2_and_ligand_batches(hip_ctx):
    """The knob on the shapes it is meant for: one full cfg2 system against the reference's sampled voxels (half of
    its atoms are hydrogens, which lose their 3.5 .. 5 A shell: a third fewer entries per tile), and a batch of cfg3
    poses through the workgroup-per-item kernel against the oracle."""
    from moleculekit_amd import batch
    g = golden("cfg2_sampled.npz")
    p = synth_config(2, 1)
    origin, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    try:
        hip_ctx.set_value_tolerance(1e-6)
        feats = batch.voxelize_lattice(p["coords"], p["atom_offsets"], p["sigmas"], origin[None], nv, p["voxelsize"], ctx=hip_ctx)[0]
        assert np.abs(feats[g["sample_idx"]] - g["sample_features"]).max() <= TOL
        q = synth_config(3, 64)
        origins = np.stack([grid_origin(c, q["boxsize"], q["voxelsize"])[0] for c in q["centers"]])
        nv3 = grid_origin(q["centers"][0], q["boxsize"], q["voxelsize"])[1]
        got = batch.voxelize_lattice(q["coords"], q["atom_offsets"], q["sigmas"], origins, nv3, q["voxelsize"], ctx=hip_ctx)
        exp = oracle_lattice(q["coords"], q["atom_offsets"], q["sigmas"], origins, nv3, q["voxelsize"])
        assert np.abs(got - exp).max() <= TOL
    finally:
        hip_ctx.set_value_tolerance(0.0)


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "voxel15", "dense_mixed", "cutoff_adversarial_1A"])
def test_direct_binning_matches_the_chain_bit_for_bit(hip_ctx, name):
    """k_bin_direct (one pass, records in place, class ids from the previous call's table) with the count / scan / fill
    chain as its device-side fallback: forced on, repeated calls on one context (the first builds the table through the
    fallback, the next go direct), against the chain alone."""
    from moleculekit_amd import batch
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    try:
        hip_ctx.set_prepass_mode(0)
        hip_ctx.set_direct_binning(0)
        ref = batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx)
        hip_ctx.set_direct_binning(1)
        for _ in range(3):
            got = batch.voxelize_lattice(*args, box=case["box"], ctx=hip_ctx)
            assert np.array_equal(got, ref)
    finally:
        hip_ctx.set_direct_binning(-1)
        hip_ctx.set_prepass_mode(-1)
    check(case, ref)


def test_direct_binning_full_size_cfg2_items(hip_ctx):
    """A call big enough for the automatic mode's threshold (8 cfg2 systems = 400 000 atoms), direct forced on: three
    calls, all equal to the chain's result; one item sampled against the oracle."""
    from moleculekit_amd import batch
    p = synth_config(2, 8, seed=2)
    origins = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    args = (p["coords"], p["atom_offsets"], p["sigmas"], origins, nv, p["voxelsize"])
    try:
        hip_ctx.set_direct_binning(0)
        ref = batch.voxelize_lattice(*args, ctx=hip_ctx)
        hip_ctx.set_direct_binning(1)
        for _ in range(3):
            assert np.array_equal(batch.voxelize_lattice(*args, ctx=hip_ctx), ref)
    finally:
        hip_ctx.set_direct_binning(-1)
    # item 3 against the oracle on 4096 of its voxels
    idx = np.random.default_rng(8).choice(int(np.prod(nv)), 4096, replace=False)
    a0, a1 = p["atom_offsets"][3], p["atom_offsets"][4]
    centers = oracle.grid_centers(origins[3], nv, p["voxelsize"])[idx]
    assert np.abs(ref[3][idx] - oracle.calculate_occupancy(centers, p["coords"][a0:a1], p["sigmas"][a0:a1])).max() <= TOL


def test_direct_binning_leaves_the_fixup_what_it_needs(hip_ctx):
    """The exact cut-off fix-up reads every atom's sigma class from the temp descriptors of the SAME call.  A context that
    voxelized another batch first (same sigma classes: the table is there, stale descriptors too) must still re-evaluate
    the wide single-sigma atoms of an adversarial batch (voxel centres at d^2 = 25 +- 1e-6 .. 1e-4 A^2) when that batch
    goes through the direct pass."""
    from moleculekit_amd import batch
    from tests.cases import case_cutoff_adversarial
    case = case_cutoff_adversarial(1.0)
    sig = np.where(case["sigmas"] != 0, case["sigmas"].max(axis=1, keepdims=True), 0.0)
    args = (case["coords"], case["atom_offsets"], sig, case["origins"], case["nvoxels"], case["voxelsize"])
    exp = oracle_lattice(*args)
    rng = np.random.default_rng(4)
    other = (rng.uniform(0, 30, case["coords"].shape).astype(np.float32), case["atom_offsets"], np.roll(sig, 37, axis=0),
             case["origins"], case["nvoxels"], case["voxelsize"])
    try:
        hip_ctx.set_prepass_mode(0)
        hip_ctx.set_direct_binning(0)
        chain = batch.voxelize_lattice(*args, ctx=hip_ctx)
        hip_ctx.set_direct_binning(1)
        for _ in range(2):
            batch.voxelize_lattice(*other, ctx=hip_ctx)                    # leaves the class table -- and ITS descriptors
        direct = batch.voxelize_lattice(*args, ctx=hip_ctx)
    finally:
        hip_ctx.set_direct_binning(-1)
        hip_ctx.set_prepass_mode(-1)
    assert np.abs(chain - exp).max() <= TOL
    assert np.array_equal(direct, chain)


# ---- the one-launch pre-pass of small calls (k_bin_solo) ---------------------------------------------------------------
def _chain_then(hip_ctx, args, box, mode, calls=1):
    from moleculekit_amd import batch
    try:
        hip_ctx.set_prepass_mode(0)
        hip_ctx.set_direct_binning(0)
        ref = batch.voxelize_lattice(*args, box=box, ctx=hip_ctx)
        hip_ctx.set_prepass_mode(-1)
        hip_ctx.set_direct_binning(mode)
        outs = [batch.voxelize_lattice(*args, box=box, ctx=hip_ctx) for _ in range(calls)]
    finally:
        hip_ctx.set_direct_binning(-1)
        hip_ctx.set_prepass_mode(-1)
    return ref, outs


@pytest.mark.parametrize("name", ["cfg1_3ptb", "ragged_batch", "voxel15", "dense_mixed", "cutoff_adversarial_1A", "special_sigmas",
                                  "tiny_items", "dense_with_wide_sigmas", "nonfinite_coords", "channels11", "pbc_batch"])
def test_solo_prepass_matches_the_chain_bit_for_bit(hip_ctx, name):
    """k_bin_solo forced on (mode 2), three calls on the module's context -- whose class table has seen whatever the
    tests before left in it -- against the count / scan / fill chain."""
    case = LATTICE_CASES[name]()
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    ref, outs = _chain_then(hip_ctx, args, case["box"], 2, calls=3)
    for got in outs:
        assert np.array_equal(got, ref)
    check(case, ref)


def test_solo_prepass_keeps_its_class_table_across_workloads(hip_ctx):
    """One context, a sequence of different molecules in the automatic mode (small calls: the one-launch pre-pass): radii
    the table has, radii it has not (inserted on the spot), 20 radii at once (overflow: general path, table emptied by
    k_tail), then the first molecule again -- each call equals the chain's result for the same input, and the oracle."""
    from moleculekit_amd import batch
    rng = np.random.default_rng(19)

    def molecule(n, radii, two=0.0):
        coords = rng.uniform(-2, 26, (n, 3)).astype(np.float32)
        r = rng.choice(radii, n)
        sig = np.zeros((n, 8))
        sig[:, 7] = r
        sig[:, 0] = r * (rng.random(n) < 0.4)
        sig[:, 3] = rng.choice(radii, n) * (rng.random(n) < two)          # a second, different radius on some atoms
        return coords, np.array([0, n]), sig, np.zeros((1, 3)), np.array([24, 24, 24]), 1.0

    seq = [molecule(900, [1.1, 1.7]), molecule(1200, [1.1, 1.52, 1.55, 1.7, 1.8]), molecule(700, [2.27, 1.7, 3.0], two=0.2),
           molecule(800, np.linspace(1.0, 2.9, 20), two=0.3), molecule(500, [1.2, 1.3])]
    seq.append(seq[0])
    try:
        hip_ctx.set_prepass_mode(0)
        hip_ctx.set_direct_binning(0)
        refs = [batch.voxelize_lattice(*m, ctx=hip_ctx) for m in seq]
        hip_ctx.set_prepass_mode(-1)
        hip_ctx.set_direct_binning(-1)
        for rep in range(2):
            for m, ref in zip(seq, refs):
                assert np.array_equal(batch.voxelize_lattice(*m, ctx=hip_ctx), ref)
    finally:
        hip_ctx.set_direct_binning(-1)
        hip_ctx.set_prepass_mode(-1)
    for m, ref in zip(seq[:4], refs):
        assert np.abs(ref - oracle_lattice(*m)).max() <= TOL


def test_one_cfg2_item_per_call_default_path(hip_ctx):
    """The 64^3 grid of one 50 000-atom system in the automatic mode (1 024 tile waves: one-launch pre-pass + the team kernel),
    repeated, against the chain and 4 096 oracle-sampled voxels."""
    from moleculekit_amd import batch
    p = synth_config(2, 1, seed=5)
    o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    args = (p["coords"], p["atom_offsets"], p["sigmas"], o[None], nv, p["voxelsize"])
    ref, outs = _chain_then(hip_ctx, args, None, -1, calls=3)
    for got in outs:
        assert np.array_equal(got, ref)
    rng = np.random.default_rng(0)
    vox = rng.choice(int(np.prod(nv)), 4096, replace=False)
    centers = np.stack(np.unravel_index(vox, nv), axis=1) * p["voxelsize"] + o
    exp = oracle.calculate_occupancy(centers, p["coords"], p["sigmas"])
    assert np.abs(ref[0][vox] - exp).max() <= TOL


def test_host_pass_is_repeated_when_the_tail_changes_values(hip_ctx):
    """A small synchronous host call takes its result out of the pinned buffer as soon as the TILE kernel is done -- while
    k_tail still runs.  When k_tail changes values (wide sigmas with voxel centres on the cut-off shell: the exact
    fix-up; tiles too dense for the LDS tier) it says so and the pass is repeated: 300 calls of the adversarial case,
    float32 and float64 results, every one equal to the first and inside the bound against the oracle."""
    from moleculekit_amd import batch
    from tests.cases import case_cutoff_adversarial
    case = case_cutoff_adversarial(1.0)
    args = (case["coords"], case["atom_offsets"], case["sigmas"], case["origins"], case["nvoxels"], case["voxelsize"])
    exp = oracle_lattice(*args)
    first = batch.voxelize_lattice(*args, ctx=hip_ctx)
    assert np.abs(first - exp).max() <= TOL
    out64 = np.empty(first.shape, np.float64)
    for i in range(300):
        if i % 2:
            got = batch.voxelize_lattice(*args, ctx=hip_ctx)
            assert np.array_equal(got, first), i
        else:
            batch.voxelize_lattice(*args, out=out64, ctx=hip_ctx)
            assert np.array_equal(out64, first.astype(np.float64)), i
    try:                                                                   # ... and with the tiles forced through the dense instance
        hip_ctx.set_lds_tier(0)
        dense = LATTICE_CASES["dense_with_wide_sigmas"]()
        dargs = (dense["coords"], dense["atom_offsets"], dense["sigmas"], dense["origins"], dense["nvoxels"], dense["voxelsize"])
        dref = batch.voxelize_lattice(*dargs, ctx=hip_ctx)
        check(dense, dref)
        for i in range(100):
            assert np.array_equal(batch.voxelize_lattice(*dargs, ctx=hip_ctx), dref), i
    finally:
        hip_ctx.set_lds_tier(-1)


def _shell_case(voxelsize, seed):
    """Atoms on lattice points (so that 30 voxels each sit at EXACTLY the 5 A cutoff: the integer triples (5,0,0), (3,4,0) and
    their permutations, scaled by the voxel size), then moved by a few float32 ulps per axis: every one of those voxels lands
    within ~1e-6 A of the shell, on either side.  One atom per 12 A cube: a shell voxel sees its own atom only, so a wrong
    cut-off decision is not hidden by a neighbour's larger value.  sigmas 1.80 (value step at the cutoff 4.7e-6, decided in
    float32) and 1.81 (5.06e-6: re-decided in double by k_tail's fix-up waves) in separate channels."""
    rng = np.random.default_rng(seed)
    n = 4
    base = np.stack(np.meshgrid(*[6.0 + 12.0 * np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    coords = base.copy()
    for k in range(len(coords)):
        for ax in range(3):
            steps = int(rng.integers(-3, 4))
            for _ in range(abs(steps)):
                coords[k, ax] = np.nextafter(coords[k, ax], np.float32(np.inf if steps > 0 else -np.inf))
    sig = np.zeros((len(coords), 8))
    sig[:, 0] = 1.80
    sig[:, 1] = 1.81
    sig[::2, 2] = 1.80
    sig[1::2, 2] = 1.81        # both classes in one channel
    sig[:, 7] = np.where(np.arange(len(coords)) % 3 == 0, 1.80, 1.7)
    nv = np.array([int(round(12.0 * n / voxelsize))] * 3)
    return coords, sig, np.zeros((1, 3)), nv


@pytest.mark.parametrize("voxelsize", [1.0, 0.5])
@pytest.mark.parametrize("tile_k", [4, 8])
def test_voxels_within_1e_6_of_the_cutoff_shell(hip_ctx, tile_k, voxelsize):
    """VERDICT r4, parity thin spot (b): the error budget at its edge (occupancy_utils.pyx:53 tests d^2 < 25 in double).  Every
    shell voxel of every atom is within ~1e-6 A of 5 A; sigma = 1.80 A is the widest sigma whose cut-off decision is left to
    float32 (a misclassified pair costs 4.74e-6), 1.81 A the narrowest that is re-decided exactly.  Both tile depths; the
    worst deviation is printed (pytest -s) and must stay within the 1e-5 bar."""
    from moleculekit_amd import batch
    worst = 0.0
    flips = 0
    for seed in (1, 2, 3):
        coords, sig, origins, nv = _shell_case(voxelsize, seed)
        offs = np.array([0, len(coords)])
        exp = oracle_lattice(coords, offs, sig, origins, nv, voxelsize)
        # the case is what it claims: per atom, voxels whose distance (in double, as the reference computes it) is within 2e-6 A of 5 A
        cen = oracle.grid_centers(origins[0], nv, voxelsize)
        d = np.sqrt(((cen[:, None, :] - coords[None, :4].astype(np.float64)) ** 2).sum(-1))
        assert (np.abs(d - 5.0) < 2e-6).sum() >= 4 * 15          # (of 30 lattice points per atom at exactly 5 A before the nudge)
        hip_ctx.set_tile_k(tile_k)
        try:
            got = batch.voxelize_lattice(coords, offs, sig, origins, nv, voxelsize, ctx=hip_ctx)
        finally:
            hip_ctx.set_tile_k(0)
        err = np.abs(got.astype(np.float64) - exp)
        worst = max(worst, float(err.max()))
        flips += int(((got[0] == 0) != (exp[0] == 0)).sum())
        # the exactly re-decided class (channel 1: sigma 1.81 only) has no cut-off error at all: what is left is float32 noise
        assert err[0, :, 1].max() <= 5e-6
        assert not ((got[0, :, 1] == 0) != (exp[0, :, 1] == 0)).any()
    print(f"cutoff shell, voxelsize {voxelsize}, K = {tile_k}: worst |gpu - reference| = {worst:.3e} (bar {TOL:g}), "
          f"{flips} (voxel, channel) values on the other side of the cutoff than the reference's")
    assert worst <= TOL
