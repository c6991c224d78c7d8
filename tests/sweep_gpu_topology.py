#!/usr/bin/env python3
"""tests/sweep_gpu_topology.py [first_seed] [count] -- random TOPOLOGY calls (frames of one molecule through a handle) whose molecule has wide sigmas
(ions: the exact cut-off fix-up) on the GPU: the split fix-up (k_exact_shells -> k_exact_redo, the default), the same call with the hits recomputed inside
k_tail (mkamd_ctx_set_exact_redo(-1)) and the PLAIN call on the sigma matrix repeated per frame must agree bit for bit.  Every seed draws its own atom
count (50 ... 7 000: up to four slices of k_exact_redo), frame count, channel count, voxel size, grid, box (periodic or not) and a palette of sigmas with
one to three wide values; a share of the wide atoms sits on lattice points nudged by float32 ulps, which puts dozens of voxels each within 1e-6 A of the
5 A shell (hundreds of hits per call instead of a handful)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
bad, hits_seen, multi = 0, 0, 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([rng.integers(50, 400), rng.integers(400, 2500), rng.integers(2500, 7000)]))
    F = int(rng.integers(1, 7))
    C = int(rng.choice([3, 8, 11]))
    vs = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
    nv = rng.integers(6, 30, size=3)
    ext = nv * vs
    pbc = bool(rng.integers(2))
    origin = rng.uniform(-20, 20, size=3)
    palette = np.concatenate([rng.uniform(1.0, 1.8, size=int(rng.integers(2, 6))), rng.uniform(1.85, 3.0, size=int(rng.integers(1, 4)))])
    sig = np.zeros((n, C), np.float32)
    for c in range(C):
        on = rng.random(n) < rng.uniform(0.1, 0.7)
        sig[on, c] = rng.choice(palette[:-1] if rng.random() < 0.5 else palette, size=int(on.sum()))
    wide_atoms = rng.choice(n, max(1, int(n * rng.choice([0.002, 0.01, 0.1]))), replace=False)
    sig[wide_atoms, int(rng.integers(C))] = palette[-1]
    base = origin + rng.uniform(-3, ext + 3, size=(n, 3))
    frames = []
    for _ in range(F):
        c = (base + rng.normal(0, 0.3, size=(n, 3))).astype(np.float32)
        for k in wide_atoms[: max(1, len(wide_atoms) // 3)]:                 # on lattice points, a few ulps off
            p = (origin + vs * np.round((c[k] - origin) / vs)).astype(np.float32)
            for ax in range(3):
                for _ in range(int(rng.integers(0, 4))):
                    p[ax] = np.nextafter(p[ax], np.float32(np.inf if rng.integers(2) else -np.inf))
            c[k] = p
        frames.append(c)
    coords = np.concatenate(frames)
    box = None
    if pbc:
        box = np.tile((ext + rng.uniform(11, 25, size=3)).astype(np.float32), (F, 1))
    d_xyz, d_offs = t(coords, np.float32), t(np.arange(F + 1) * n, np.int64)
    d_org = t(np.tile(origin, (F, 1)), np.float64)
    d_box = None if box is None else t(box, np.float32)
    d_sig1 = t(sig, np.float32)
    try:
        topo = _lib.Topology(ctx, d_sig1, vs)
    except ValueError as e:                                                   # (more than 15 distinct sigmas in a channel group: not a topology's case)
        print(f"seed {seed}: skipped ({e})", flush=True)
        continue
    plain = batch.voxelize_lattice_torch(d_xyz, d_offs, d_sig1.repeat(F, 1).contiguous(), d_org, nv, vs, box=d_box, ctx=ctx)
    ctx.set_exact_redo(0)
    split = batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, vs, box=d_box, ctx=ctx, topology=topo)
    ctx.set_exact_redo(-1)
    inside = batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, vs, box=d_box, ctx=ctx, topology=topo)
    ctx.set_exact_redo(0)
    ctx.synchronize()
    ok = torch.equal(plain, split) and torch.equal(plain, inside)
    multi += n > 2048
    if not ok:
        bad += 1
        print(f"seed {seed}: MISMATCH n {n} F {F} C {C} vs {vs} nv {nv.tolist()} pbc {pbc}: split != plain at {int((plain != split).sum())} values, "
              f"inside != plain at {int((plain != inside).sum())}", flush=True)
    topo.close()
print(f"topology sweep, seeds {first}..{first + count - 1}: {count - bad} of {count} calls bit-identical (split fix-up == inside k_tail == plain call); "
      f"{multi} molecules of more than one k_exact_redo slice", flush=True)
sys.exit(1 if bad else 0)
