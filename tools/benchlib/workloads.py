"""The bench workloads: BASELINE.json's configurations as synthetic inputs, their algorithmic bytes / flops."""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP32_VALU_PEAK_TFLOPS = 157.3  # vector FP32 peak (secondary roofline: cfg2/cfg4 are VALU-bound)

# items per GPU per step: big enough that the tile grid is many waves deep (a 32-grid step of cfg2 is only four:
# its tail costs ~15 %) and that the fixed per-step costs (launch gaps, the barrier of the timed region) amortise
# (cfg5's nominal batch is 100 000 items, BASELINE.json configs[4]; the small-molecule workloads keep gaining up to
#  ~32 k grids per step: DESIGN.md section 5)
DEFAULT_BATCH = {"cfg1": 4096, "cfg2": 256, "cfg3": 32768, "cfg4": 256, "cfg5": 65536, "dist": 2048, "dropin": 1}



def real_protein_config(batch, seed):
    """BASELINE configs[0] scaled up: the reference's own 3PTB pocket case (real protein density and
    channel typing, tests/golden/cfg1_3ptb.npz), `batch` randomly rotated copies (rotation about the
    grid centre, what the reference's augmentation loop feeds getVoxelDescriptors)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_3ptb.npz"))
    rng = np.random.default_rng(seed)
    c0 = g["coords"].astype(np.float64) - g["center"][None, :]
    q = rng.normal(size=(batch, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(batch, 3, 3)
    coords = (np.einsum("bij,nj->bni", R, c0) + g["center"][None, None, :]).astype(np.float32)
    n = c0.shape[0]
    return dict(coords=coords.reshape(-1, 3), sigmas=np.tile(g["sigmas"], (batch, 1)),
                atom_offsets=np.arange(batch + 1, dtype=np.int64) * n,
                centers=np.tile(g["center"][None, :], (batch, 1)).astype(np.float64),
                boxsize=np.asarray(g["boxsize"], dtype=np.float64), voxelsize=float(g["voxelsize"]), box=None)


def make_config(name, batch, seed=None):
    from tests.synth import synth_config
    if name == "cfg1":
        return real_protein_config(batch, 1 if seed is None else seed)
    return synth_config(int(name[3:]), batch, seed=seed)


def make_workload(name, batch, seed):
    from tests.synth import grid_origin
    p = make_config(name, batch, seed)
    if os.environ.get("MKAMD_SPATIAL_ORDER", "0") == "1":
        # experiment knob: atoms of every item in spatially coherent order (8 A blocks), like residues / waters in a
        # real topology, instead of the synthetic generator's random order
        co, sg, offs = p["coords"].copy(), p["sigmas"].copy(), p["atom_offsets"]
        for b in range(len(offs) - 1):
            s0, e0 = int(offs[b]), int(offs[b + 1])
            key = np.floor(co[s0:e0] / 8.0).astype(np.int64)
            order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
            co[s0:e0], sg[s0:e0] = co[s0:e0][order], sg[s0:e0][order]
        p["coords"], p["sigmas"] = co, sg
    origins = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    return p, origins, nv


def in_range_pairs(p, nv, items):
    """(voxel, atom x channel) pairs within the 5 A cutoff -- what the reference's loop accepts (occupancy_utils.pyx:53) --
    counted exactly on `items` of the workload (host, numpy): every atom against the lattice points of the 11^3 voxels
    around it (periodic items: its images within reach of the grid)."""
    from tests.synth import grid_origin
    vs = float(p["voxelsize"])
    R = 5.0 / vs
    w = int(np.ceil(R)) + 1
    off = np.stack(np.meshgrid(*[np.arange(-w, w + 1)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    total = 0
    for b in items:
        s, e = int(p["atom_offsets"][b]), int(p["atom_offsets"][b + 1])
        o, _ = grid_origin(p["centers"][b], p["boxsize"], vs)
        x = (p["coords"][s:e].astype(np.float64) - o) / vs                      # voxel units, voxel i at i
        nch = (np.asarray(p["sigmas"][s:e]) != 0).sum(1).astype(np.int64)
        if p["box"] is not None:                                                # images that can reach the grid
            L = p["box"][b].astype(np.float64) / vs
            sh = np.stack(np.meshgrid(*[np.arange(-1, 2)] * 3, indexing="ij"), -1).reshape(-1, 3) * L
            x = (x[None] + sh[:, None]).reshape(-1, 3)
            nch = np.tile(nch, 27)
        keep = np.all((x > -R) & (x < np.asarray(nv) - 1 + R), axis=1) & (nch > 0)
        x, nch = x[keep], nch[keep]
        for i in range(0, len(x), 4096):
            xi = x[i:i + 4096]
            base = np.rint(xi)
            pts = base[:, None, :] + off[None]                                  # [n, (2w+1)^3, 3]
            ok = (((pts - xi[:, None]) ** 2).sum(-1) < R * R) & np.all((pts >= 0) & (pts < np.asarray(nv)), axis=-1)
            total += int((ok.sum(1) * nch[i:i + 4096]).sum())
    return total


def algorithmic_flops(p, nv, C=8):
    """SURVEY.md section 8d, secondary roofline: ~22 lane-operations per in-range (voxel, entry) pair (with the min-q
    shortcut) + ~14 per voxel-channel of epilogue (12 operations, 2 transcendentals); the pairs counted on the first items
    of the batch (all of a cfg2 / cfg4 item; 16 small molecules) and scaled to the batch."""
    B = len(p["atom_offsets"]) - 1
    V = int(np.prod(nv))
    items = list(range(min(B, 1 if int(p["atom_offsets"][1]) > 5000 else 16)))
    pairs = in_range_pairs(p, nv, items) / len(items)
    return int(B * (22.0 * pairs + 14.0 * V * C)), pairs / V


def algorithmic_bytes(p, nv, C=8):
    """per grid V*C*4 (one float32 write per voxel-channel) + N*(12 + 4*C)
    (coords + per-channel sigmas read once); summed over the batch."""
    B = len(p["atom_offsets"]) - 1
    V = int(np.prod(nv))
    return B * V * C * 4 + int(p["atom_offsets"][-1]) * (12 + 4 * C)


def reduction_workload(G=200, A=15, F=512, L=60.0, seed=5):
    """tools/bench_reduction.py's protein-like trajectory (the shape the 2.9 ms of round 2 were measured on): G residues of A
    atoms (centres uniform in an L^3 box, atoms N(0, 1.5 A) around them, N(0, 0.3 A) per frame), 4 chains of G/4 residues."""
    rng = np.random.default_rng(seed)
    N = G * A
    centres = rng.uniform(0, L, size=(G, 3))
    c0 = (np.repeat(centres, A, axis=0) + rng.normal(0, 1.5, size=(N, 3))).astype(np.float32)
    coords = np.ascontiguousarray((c0[:, :, None] + rng.normal(0, 0.3, size=(N, 3, F))).astype(np.float32))
    box = np.full((3, F), L, dtype=np.float32)
    atoms = np.arange(N, dtype=np.int32)
    offs = (np.arange(G + 1, dtype=np.int64) * A)
    chains = (np.arange(G) // max(1, G // 4)).astype(np.uint32)
    return coords, box, atoms, offs, chains, np.ones(N, np.float32)
