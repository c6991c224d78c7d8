#!/usr/bin/env python3
"""tools/team_probe.py -- the 3PTB pocket and the cfg2 item, one grid per call, for every team size (waves per tile)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch
from tests.synth import grid_origin, synth_config
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
def probe(name, p, teams, reps=300):
    o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    n = int(p["atom_offsets"][1])
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    args = (t(p["coords"][:n], np.float32), t(p["atom_offsets"][:2], np.int64), t(p["sigmas"][:n], np.float32), t(o[None], np.float64), nv, p["voxelsize"])
    out = torch.empty((1, int(np.prod(nv)), 8), dtype=torch.float32, device=dev)
    ref = None
    for team in teams:
        ctx.set_tile_team(team)
        for _ in range(20):
            batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
        torch.cuda.synchronize(dev)
        us = (time.perf_counter() - t0) / reps * 1e6
        h = out.cpu().numpy().copy()
        ref = h if ref is None else ref
        print(f"{name:5s} team={team:2d}: {us:6.1f} us back to back   bitwise == first: {np.array_equal(h, ref)}", flush=True)
    ctx.set_tile_team(-1)
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cfg1_3ptb.npz"))
probe("3ptb", dict(coords=g["coords"], sigmas=g["sigmas"], atom_offsets=np.array([0, len(g["coords"])]), centers=g["center"][None],
                   boxsize=g["boxsize"], voxelsize=float(g["voxelsize"])), (4, 8, 16, -1))
probe("cfg2", synth_config(2, 1), (4, 8, -1))
