"""One-off wider sweep of tests/test_gpu_random.py's generator (GPU box; not collected by pytest): seeds beyond the
40 of the test tier, plus a variant whose atoms carry a DIFFERENT sigma per channel (the binning's multi-sigma path).
Prints the worst absolute error against the oracle.   python tests/sweep_gpu_random.py [first_seed] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_random import _config
from tests.cases import TOL, oracle_lattice
from moleculekit_amd import batch, _lib

first = int(sys.argv[1]) if len(sys.argv) > 1 else 40
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = _lib.default_context(0)
if os.environ.get("MKAMD_TILE_ITEMS") == "1":       # the workgroup-per-item kernel on every configuration
    ctx.set_tile_team(0); ctx.set_tile_items(1)
worst, where = 0.0, None
for seed in range(first, first + count):
    k = _config(seed)
    for multi in (False, True):
        sig = k["sigmas"]
        if multi:
            rng = np.random.default_rng(seed)
            sig = np.where(sig > 0, rng.choice([0.8, 1.1, 1.52, 1.7, 2.0, 2.3], size=sig.shape), 0.0)
        args = (k["coords"], k["atom_offsets"], sig, k["origins"], k["nvoxels"], k["voxelsize"])
        got = batch.voxelize_lattice(*args, box=k["box"], ctx=ctx)
        want = oracle_lattice(k["coords"], k["atom_offsets"], sig.astype(np.float64), k["origins"], k["nvoxels"],
                              k["voxelsize"], k["box"])
        err = float(np.abs(got - want).max(initial=0.0))
        if err > worst:
            worst, where = err, (seed, multi, k["voxelsize"], tuple(int(v) for v in k["nvoxels"]), k["box"] is not None)
        if err > TOL:
            print("FAIL seed", seed, "multi", multi, "err", err, "vs", k["voxelsize"])
print("seeds", first, "..", first + count - 1, "worst abs err", worst, "(tolerance", TOL, ") at (seed, multi, voxelsize, nvoxels, pbc) =", where)
