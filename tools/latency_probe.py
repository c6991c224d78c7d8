#!/usr/bin/env python3
"""tools/latency_probe.py -- latency of ONE grid per call (the reference's usage pattern) on the GPU box:
a cfg2 item (50k atoms, 64^3) and the 3PTB pocket (24^3), device-resident inputs, one synchronous call at a time,
for every (tile K, waves per tile) combination; then the drop-in getVoxelDescriptors call (host arrays in/out)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch
from tests.synth import grid_origin, synth_config

dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)


def probe(name, p, reps=50):
    o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    n = int(p["atom_offsets"][1])
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    args = (t(p["coords"][:n], np.float32), t(p["atom_offsets"][:2], np.int64), t(p["sigmas"][:n], np.float32), t(o[None], np.float64), nv, p["voxelsize"])
    out = torch.empty((1, int(np.prod(nv)), 8), dtype=torch.float32, device=dev)
    ref = None
    for k in (0, 4, 8):
        for team in (0, 1):
            ctx.set_tile_k(k); ctx.set_tile_team(team)
            for _ in range(5):
                batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
                torch.cuda.synchronize(dev)
            us = (time.perf_counter() - t0) / reps * 1e6
            t0 = time.perf_counter()
            for _ in range(reps):
                batch.voxelize_lattice_torch(*args, out=out, ctx=ctx)
            torch.cuda.synchronize(dev)
            us_b2b = (time.perf_counter() - t0) / reps * 1e6
            h = out.cpu().numpy().copy()
            if ref is None:
                ref = h
            same = np.array_equal(h, ref) if k in (0, 4) or True else True
            print(f"{name:6s} K={k} team={team}: {us:7.1f} us per synchronous call, {us_b2b:7.1f} us back to back   bitwise==first: {np.array_equal(h, ref)}", flush=True)
    ctx.set_tile_k(0); ctx.set_tile_team(-1)


probe("cfg2", synth_config(2, 1))
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cfg1_3ptb.npz"))
probe("3ptb", dict(coords=g["coords"], sigmas=g["sigmas"], atom_offsets=np.array([0, len(g["coords"])]), centers=g["center"][None],
                   boxsize=g["boxsize"], voxelsize=float(g["voxelsize"])))
from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
kw = dict(boxsize=[24, 24, 24], center=g["center"], voxelsize=1, usercoords=g["coords"], userchannels=g["sigmas"])
for team in (0, -1):
    ctx.set_tile_team(team)
    for _ in range(5):
        getVoxelDescriptors(None, **kw)
    t0 = time.perf_counter()
    for _ in range(100):
        f, c, n = getVoxelDescriptors(None, **kw)
    print(f"drop-in getVoxelDescriptors(3PTB) team={team}: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms per call, max err {np.abs(f - g['features']).max():.2e}")
ctx.set_tile_team(-1)


def probe_usercenters():
    """The drop-in call when the caller passes the centres (`usercenters`, the ML pipelines' usage: centres computed
    once, coordinates rotated per sample): a lattice is recognised on the host, anything else goes to the pairwise kernel."""
    from moleculekit_amd.voxeldescriptors import getCenters
    coords, chans = g["coords"], g["sigmas"]
    centers, nvox = getCenters(boxsize=[24, 24, 24], center=g["center"], voxelsize=1)
    rng = np.random.default_rng(0)
    jitter = centers + rng.normal(0, 1e-3, centers.shape)                     # not a lattice any more
    shifted = [centers + np.array([0.01 * k, 0.0, 0.0]) for k in range(8)]   # a different lattice every call (translation augmentation)
    for name, c in (("lattice usercenters", centers), ("lattice usercenters, shifted every call", shifted), ("arbitrary usercenters", jitter)):
        pick = (lambda k: c[k % len(c)]) if isinstance(c, list) else (lambda k: c)
        for k in range(20):
            getVoxelDescriptors(None, usercenters=pick(k), userchannels=chans, usercoords=coords)
        t0 = time.perf_counter(); n = 200
        for k in range(n):
            f, _c = getVoxelDescriptors(None, usercenters=pick(k), userchannels=chans, usercoords=coords)
        err = np.abs(f - g["features"]).max() if c is centers else float("nan")
        print(f"drop-in getVoxelDescriptors(3PTB, {name}): {(time.perf_counter() - t0) / n * 1e3:.4f} ms per call, max err vs golden {err:.2e}")


probe_usercenters()
