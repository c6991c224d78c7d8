#!/usr/bin/env python3
"""tools/bench_voxelize_trajectory.py -- voxelizeTrajectory (host arrays in, host features out) on the cfg4 shape:
30 000 atoms x 256 frames, 48^3 grid, periodic.  MKAMD_LIB selects the build."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moleculekit_amd import batch
rng = np.random.default_rng(4)
N, F, L = 30000, 256, 66.9
c0 = rng.uniform(0, L, size=(N, 3)).astype(np.float32)
coords = np.ascontiguousarray((c0[:, :, None] + rng.normal(0, 0.3, size=(N, 3, F))).astype(np.float32) % np.float32(L))
box = np.full((3, F), L, np.float32)
rad = rng.choice([1.1, 1.7, 1.55, 1.52, 1.8], size=N, p=[.5, .3, .08, .11, .01])
mask = rng.random((N, 8)) < np.array([.3, .05, .1, .05, .02, .02, .001, 1.0]); mask[:, 7] = rad != 1.1
sig = rad[:, None] * mask
center = np.array([L / 2] * 3)
for _ in range(2):
    f, o, nv = batch.voxelizeTrajectory(coords, sig, center, [48, 48, 48], 1.0, box=box)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); f, o, nv = batch.voxelizeTrajectory(coords, sig, center, [48, 48, 48], 1.0, box=box); ts.append(time.perf_counter() - t0)
print(f"voxelizeTrajectory: {min(ts) * 1e3:.1f} ms for {F} frames ({f.nbytes / 1e6:.0f} MB of features): {F / min(ts):.0f} frames/s, checksum {float(f[::37].sum()):.6e}")
import torch
acc = torch.zeros((), device="cuda", dtype=torch.float64)
def it():
    for idx, feats in batch.iterVoxelizeTrajectory(coords, sig, center, [48, 48, 48], 1.0, box=box, chunk=128):
        acc.add_(feats[::37].sum(dtype=torch.float64))
    torch.cuda.synchronize()
it(); t0 = time.perf_counter(); it(); dt = time.perf_counter() - t0
print(f"iterVoxelizeTrajectory (features stay on the device): {dt * 1e3:.1f} ms: {F / dt:.0f} frames/s")
