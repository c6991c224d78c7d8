#!/usr/bin/env python3
"""tools/bench_host_batch.py -- the batched call with HOST arrays in and out (PCIe both ways, never bench.py's `value`):
64 cfg2 grids per call, the result into a FRESH array each time (what getVoxelDescriptorsBatch returns) and into a
reused one.  MKAMD_LIB selects the build."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moleculekit_amd import batch
from tests.synth import synth_config, grid_origin

B = 64
p = synth_config(2, B)
orgs = np.stack([grid_origin(p["centers"][b], p["boxsize"], p["voxelsize"])[0] for b in range(B)])
nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
args = (p["coords"], p["atom_offsets"], p["sigmas"].astype(np.float32), orgs, nv, p["voxelsize"])
V = int(np.prod(nv))
batch.voxelize_lattice(*args)
for name, reuse in (("fresh result array", False), ("reused result array", True)):
    out = np.empty((B, V, 8), np.float32) if reuse else None
    if reuse: out[:] = 0
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); r = batch.voxelize_lattice(*args, out=out); ts.append(time.perf_counter() - t0); del r
    print(f"{name}: {min(ts) * 1e3:.1f} ms per call of {B} grids ({B * V * 8 * 4 / 1e6:.0f} MB out): {B * V * 8 / min(ts) / 1e9:.2f} G voxel-channels/s, {B * V * 8 * 4 / min(ts) / 1e9:.1f} GB/s of results")
