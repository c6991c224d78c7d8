#!/usr/bin/env python3
"""tools/team_probe2.py -- one cfg2 grid per call: tile depth x team size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch
from tests.synth import grid_origin, synth_config
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
p = synth_config(2, 1)
o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
n = int(p["atom_offsets"][1])
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
args = (t(p["coords"][:n], np.float32), t(p["atom_offsets"][:2], np.int64), t(p["sigmas"][:n], np.float32), t(o[None], np.float64), nv, p["voxelsize"])
out = torch.empty((1, int(np.prod(nv)), 8), dtype=torch.float32, device=dev)
for rep in range(2):
    for k, team in ((4, 4), (8, 4), (4, -1), (0, -1)):
        ctx.set_tile_k(k); ctx.set_tile_team(team)
        for _ in range(20):
            batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(300):
            batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
        torch.cuda.synchronize(dev)
        print(f"K={k} team={team:2d}: {(time.perf_counter() - t0) / 300 * 1e6:6.1f} us", flush=True)
ctx.set_tile_k(0); ctx.set_tile_team(-1)
