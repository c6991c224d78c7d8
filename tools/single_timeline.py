#!/usr/bin/env python3
"""tools/single_timeline.py -- ONE cfg2 grid (50k atoms, 64^3) per synchronous call, 30 calls; run under
`rocprofv3 --kernel-trace` and summarised by tools/single_timeline_report.py: where the ~70 us of a call go."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch
from tests.synth import grid_origin, synth_config

dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if which == "cfg2":
    p = synth_config(2, 1)
else:
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cfg1_3ptb.npz"))
    p = dict(coords=g["coords"], sigmas=g["sigmas"], atom_offsets=np.array([0, len(g["coords"])]), centers=g["center"][None],
             boxsize=g["boxsize"], voxelsize=float(g["voxelsize"]))
o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
n = int(p["atom_offsets"][1])
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
args = (t(p["coords"][:n], np.float32), t(p["atom_offsets"][:2], np.int64), t(p["sigmas"][:n], np.float32), t(o[None], np.float64), nv, p["voxelsize"])
out = torch.empty((1, int(np.prod(nv)), 8), dtype=torch.float32, device=dev)
for _ in range(30):
    batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
    torch.cuda.synchronize(dev)
