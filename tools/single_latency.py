#!/usr/bin/env python3
"""tools/single_latency.py -- one grid per call, automatic mode, device-resident inputs: microseconds per call back to back
(what bench.py reports as single_grid_latency_us) for the cfg2 item and the 3PTB pocket, and the drop-in getVoxelDescriptors
call.  MKAMD_LIB selects the library build (A/B runs on one box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch
from tests.synth import grid_origin, synth_config

dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
if os.environ.get("MKAMD_TILE_K"):                     # A/B: force the tile depth (4 / 8) of every call
    ctx.set_tile_k(int(os.environ["MKAMD_TILE_K"]))
    _lib.default_context().set_tile_k(int(os.environ["MKAMD_TILE_K"]))


def probe(p, reps=300):
    o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    n = int(p["atom_offsets"][1])
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    args = (t(p["coords"][:n], np.float32), t(p["atom_offsets"][:2], np.int64), t(p["sigmas"][:n], np.float32), t(o[None], np.float64), nv, p["voxelsize"])
    out = torch.empty((1, int(np.prod(nv)), 8), dtype=torch.float32, device=dev)
    best = 1e9
    for _ in range(3):
        for _ in range(20):
            batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
        torch.cuda.synchronize(dev)
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cfg1_3ptb.npz"))
a = probe(synth_config(2, 1))
b = probe(dict(coords=g["coords"], sigmas=g["sigmas"], atom_offsets=np.array([0, len(g["coords"])]), centers=g["center"][None],
               boxsize=g["boxsize"], voxelsize=float(g["voxelsize"])))
from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
kw = dict(boxsize=[24, 24, 24], center=g["center"], voxelsize=1, usercoords=g["coords"], userchannels=g["sigmas"])
best = 1e9
for _ in range(3):
    for _ in range(10):
        getVoxelDescriptors(None, **kw)
    t0 = time.perf_counter()
    for _ in range(200):
        getVoxelDescriptors(None, **kw)
    best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
print(f"{os.environ.get('MKAMD_LIB', 'libmkamd.so') + ' K=' + os.environ.get('MKAMD_TILE_K', 'auto'):40s} cfg2 grid {a:6.1f} us   3PTB grid {b:6.1f} us   drop-in 3PTB call {best:.4f} ms", flush=True)
