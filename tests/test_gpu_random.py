"""GPU tier: seeded random configurations through every execution variant of the lattice path.

For each configuration (batch size, ragged atom counts incl. empty items, density from sparse to far denser
than any LDS tier, voxel size, odd grid shapes, 1..12 channels, few / many sigma classes, optional periodic
boxes) the result must (a) stay within TOL of the oracle and (b) for a given tile depth K be BIT-IDENTICAL
whatever the LDS tier, the class-sorted / general / dense path and the pre-pass (kernel chain / one launch per
item) -- those variants only change how the same
minima are scheduled.  (K itself moves the tile centre the coordinates are made relative to, i.e. the
rounding of the last bit: K = 4 and K = 8 agree to ~1e-7, not bitwise.)"""
import numpy as np
import pytest

from tests.cases import TOL, oracle_lattice

pytestmark = pytest.mark.gpu


def _config(seed):
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 9))
    vs = float(rng.choice([0.5, 0.7, 1.0, 1.0, 1.5, 2.0]))
    nv = rng.integers(3, 21, size=3)
    C = int(rng.choice([1, 3, 8, 8, 8, 12]))
    extent = nv * vs
    density = float(rng.choice([0.002, 0.02, 0.1, 0.1, 0.4]))            # atoms / A^3 (0.4: tiles over every tier)
    span = extent + 12.0
    nmean = min(int(density * np.prod(span)), 6000)
    ns = [int(n) for n in rng.integers(0, 2 * nmean + 2, size=B)]
    if seed % 3 == 0:
        ns[int(rng.integers(0, B))] = 0
    pbc = seed % 4 == 1
    origins = rng.uniform(-5, 5, size=(B, 3))
    coords, box = [], None
    if pbc:
        box = np.maximum(rng.uniform(0.6, 1.4, size=(B, 3)) * span, 10.5).astype(np.float32)
    for b, n in enumerate(ns):
        lo = origins[b] - 6.0
        c = lo + rng.random((n, 3)) * span
        if pbc:
            c += rng.integers(-2, 3, size=(n, 3)) * box[b]                # unwrapped coordinates
        coords.append(c.astype(np.float32))
    n_classes = int(rng.choice([1, 3, 5, 40]))                            # 40 > 15: general path for the whole call
    radii = rng.uniform(0.9, 2.3, size=n_classes)
    N = sum(ns)
    sig = radii[rng.integers(0, n_classes, size=(N, 1))] * (rng.random((N, C)) < rng.uniform(0.1, 0.9))
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    cc = np.concatenate(coords) if N else np.zeros((0, 3), np.float32)
    return dict(coords=cc, atom_offsets=offs, sigmas=sig, origins=origins, nvoxels=nv.astype(np.int64), voxelsize=vs, box=box)


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration_all_variants(hip_ctx, seed):
    from moleculekit_amd import batch
    k = _config(seed)
    args = (k["coords"], k["atom_offsets"], k["sigmas"], k["origins"], k["nvoxels"], k["voxelsize"])
    base = batch.voxelize_lattice(*args, box=k["box"], ctx=hip_ctx)
    want = oracle_lattice(k["coords"], k["atom_offsets"], k["sigmas"].astype(np.float64), k["origins"], k["nvoxels"],
                          k["voxelsize"], k["box"])
    assert base.shape == want.shape
    assert np.abs(base - want).max(initial=0.0) <= TOL
    try:
        for tile_k in (4, 8):
            ref = None
            for tier, general, prepass in [(0, False, 0), (1, False, 1), (2, False, 0), (-1, False, 1), (-1, True, 0), (-1, True, 1)]:
                hip_ctx.set_tile_k(tile_k); hip_ctx.set_lds_tier(tier); hip_ctx.set_force_general(general)
                hip_ctx.set_prepass_mode(prepass)
                got = batch.voxelize_lattice(*args, box=k["box"], ctx=hip_ctx)
                if ref is None:
                    ref = got
                    assert np.abs(got - want).max(initial=0.0) <= TOL
                    assert np.abs(got - base).max(initial=0.0) <= 6e-6   # another K = other plane constants (and another fast/exact split of the classes)
                else:
                    assert np.array_equal(got, ref), (tile_k, tier, general, prepass)
            # a workgroup per item (what batches of ligand-sized items get): sorted once per item, or -- items of more
            # than 384 entries, too many classes, the forced general path -- walked unsorted; the same bits
            for general, prepass in ((False, 1), (False, 0), (True, 1)):
                hip_ctx.set_tile_k(tile_k); hip_ctx.set_lds_tier(-1); hip_ctx.set_force_general(general)
                hip_ctx.set_prepass_mode(prepass); hip_ctx.set_tile_team(0); hip_ctx.set_tile_items(1)
                got = batch.voxelize_lattice(*args, box=k["box"], ctx=hip_ctx)
                hip_ctx.set_tile_team(-1); hip_ctx.set_tile_items(-1)
                assert np.array_equal(got, ref), (tile_k, "items", general, prepass)
    finally:
        hip_ctx.set_tile_k(0); hip_ctx.set_lds_tier(-1); hip_ctx.set_force_general(False); hip_ctx.set_prepass_mode(-1)
        hip_ctx.set_tile_team(-1); hip_ctx.set_tile_items(-1)
