"""tools/path_identity_probe.py -- is a frame's result the same bits whatever the call it is part of?  One cfg4 frame voxelized in
calls of 1, 2 and 16 items, automatic mode and with every path knob forced; float32 and float64 sigmas."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from moleculekit_amd import batch, _lib
ctx = _lib.default_context(0)
p, _, _ = bench.make_workload("cfg4", 16, seed=4004)
N = int(p["atom_offsets"][1])
c = p["coords"].reshape(16, N, 3)
nv = np.ceil(p["boxsize"] / p["voxelsize"]).astype(int)
org = (p["centers"] - p["boxsize"] / 2)
print("origins identical:", bool((org == org[0]).all()), "boxes identical:", bool((p["box"] == p["box"][0]).all()))
def vox(frames, sig, pbc=True):
    B = len(frames)
    co = np.ascontiguousarray(c[frames].reshape(-1, 3)); offs = np.arange(B + 1, dtype=np.int64) * N
    return batch.voxelize_lattice(co, offs, np.tile(sig, (B, 1)), org[:B], nv, float(p["voxelsize"]), box=p["box"][:B] if pbc else None, ctx=ctx)
for dt in (np.float32, np.float64):
    sig = np.ascontiguousarray(p["sigmas"][:N], dtype=dt)
    ref = vox([15], sig)[0]
    def same(name, B, **knobs):
        for k, v in knobs.items(): getattr(ctx, "set_" + k)(v)
        r = vox(list(range(16 - B, 16)), sig)[B - 1]
        for k in knobs: getattr(ctx, "set_" + k)(-1)
        print(f"  {dt.__name__} B={B:2d} {name:34s}: identical to the one-item call: {np.array_equal(r, ref)}   max |diff| {float(np.abs(r - ref).max()):.2e}")
    same("automatic", 2); same("automatic", 16)
    same("direct_binning=0 (chain)", 2, direct_binning=0); same("direct_binning=0 (chain)", 16, direct_binning=0)
    same("direct_binning=1", 16, direct_binning=1); same("direct_binning=2 (one launch)", 16, direct_binning=2)
    same("tile_team=0", 2, tile_team=0); same("tile_team=0", 16, tile_team=0); same("tile_team=1", 16, tile_team=1)
    same("tile_items=1", 16, tile_items=1)
