import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from moleculekit_amd import _lib, batch
ctx = _lib.default_context(0)
dev = torch.device("cuda", 0)
for wl, B in (("cfg5", 16384), ("cfg3", 8192)):
    p, origins, nv = bench.make_workload(wl, B, seed=5)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    args = (t(p["coords"], np.float32), t(p["atom_offsets"], np.int64), t(p["sigmas"], np.float32), t(origins, np.float64), nv, p["voxelsize"])
    ctx.set_tile_items(0); a = batch.voxelize_lattice_torch(*args, ctx=ctx).clone()
    ctx.set_tile_items(-1); b = batch.voxelize_lattice_torch(*args, ctx=ctx)
    torch.cuda.synchronize()
    print(wl, B, "bitwise equal:", bool(torch.equal(a, b)), "sum", float(b.double().sum()))
