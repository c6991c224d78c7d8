"""Pipelined step time when the pre-pass outweighs the tile kernel: cfg2's atoms (50 k per item, 79 A cube) voxelized
onto a SMALL grid in the middle (most atoms are binned only to be dropped).  python tools/bench_prepass_heavy.py [n]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from moleculekit_amd import _lib, batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = 128
p = bench.make_config("cfg2", B, seed=7)
dev = torch.device("cuda", 0)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
nv = np.array([n, n, n], np.int32)
origins = np.tile(np.asarray(p["centers"][0]) - 0.5 * n, (B, 1))
d = (t(p["coords"], np.float32), t(p["atom_offsets"], np.int64), t(p["sigmas"], np.float32), t(origins, np.float64))
out = torch.empty((B, n * n * n, 8), dtype=torch.float32, device=dev)
ctx = _lib.default_context(0)
for pipe in (False, True):
    ctx.set_pipelining(pipe)
    step = lambda: batch.voxelize_lattice_torch(*d, nv, 1.0, out=out, ctx=ctx)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print(f"grid {n}^3 x {B} items, pipelined={pipe}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms/step")
