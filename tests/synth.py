"""Seeded synthetic inputs for the BASELINE.json configs.

Shared by tests/golden/make_golden.py (reference side), the parity tests and bench.py, so the
GPU path, the oracle and the real reference all see byte-identical inputs.  Pure numpy; all
randomness from ``np.random.default_rng(seed)``.
"""
from __future__ import annotations

import numpy as np

_RADII = np.array([1.1, 1.7, 1.55, 1.52, 1.8])
_RADII_P = np.array([0.5, 0.3, 0.08, 0.11, 0.01])
_MASK_P = np.array([0.3, 0.05, 0.1, 0.05, 0.02, 0.02, 0.001])


def synth_sigmas(rng: np.random.Generator, n: int) -> np.ndarray:
    """(n, 8) float64 sigma matrix: radius in {H,C,N,O,S} vdW radii, channels 0-6 Bernoulli
    masks, channel 7 = (radius != 1.1)  -- i.e. `occupancies` = heavy atoms."""
    radius = rng.choice(_RADII, size=n, p=_RADII_P)
    mask = np.zeros((n, 8), dtype=np.float64)
    mask[:, :7] = rng.random((n, 7)) < _MASK_P
    mask[:, 7] = radius != 1.1
    return radius[:, None] * mask


def synth_config(cfg: int, batch: int, seed: int | None = None) -> dict:
    """Inputs of BASELINE.json config ``cfg`` (2..5), ``batch`` independent items.

    Returns dict(coords f32 [sumN,3], sigmas f64 [sumN,8], atom_offsets i64 [B+1],
    centers f64 [B,3] (grid centre per item), boxsize f64 [3], voxelsize float,
    box f32 [B,3] or None)."""
    rng = np.random.default_rng(cfg if seed is None else seed)
    box = None
    if cfg == 2:      # solvated protein: 50k atoms uniform, density 0.1 / A^3, 64^3 @ 1 A
        L = 79.37
        ns = [50000] * batch
        coords = [rng.uniform(0, L, size=(n, 3)).astype(np.float32) for n in ns]
        centers = np.full((batch, 3), 39.685)
        boxsize, vs = np.array([64.0, 64.0, 64.0]), 1.0
    elif cfg == 3:    # ligand poses: 60 atoms N(0, 3 A), 24^3 @ 1 A
        ns = [60] * batch
        coords = [rng.normal(0, 3.0, size=(n, 3)).astype(np.float32) for n in ns]
        centers = np.zeros((batch, 3))
        boxsize, vs = np.array([24.0, 24.0, 24.0]), 1.0
    elif cfg == 4:    # MD frames: 30k atoms, cubic box 66.9 A, 48^3 @ 1 A, PBC, random walk frames
        L = 66.9
        n = 30000
        ns = [n] * batch
        x = rng.uniform(0, L, size=(n, 3))
        coords = []
        for _ in range(batch):
            coords.append(x.astype(np.float32))
            x = np.mod(x + rng.normal(0, 0.3, size=(n, 3)), L)
        centers = np.full((batch, 3), L / 2)
        boxsize, vs = np.array([48.0, 48.0, 48.0]), 1.0
        box = np.full((batch, 3), L, dtype=np.float32)
    elif cfg == 5:    # virtual screen: 20..50 atoms N(0, 2 A), 12 A box @ 0.5 A -> 24^3
        ns = list(rng.integers(20, 51, size=batch))
        coords = [rng.normal(0, 2.0, size=(n, 3)).astype(np.float32) for n in ns]
        centers = np.zeros((batch, 3))
        boxsize, vs = np.array([12.0, 12.0, 12.0]), 0.5
    else:
        raise ValueError(cfg)
    if cfg == 4:      # same atoms in every frame -> same sigmas
        s1 = synth_sigmas(rng, ns[0])
        sigmas = np.concatenate([s1] * batch)
    else:
        sigmas = np.concatenate([synth_sigmas(rng, n) for n in ns])
    offs = np.zeros(batch + 1, dtype=np.int64)
    offs[1:] = np.cumsum(ns)
    return dict(coords=np.concatenate(coords), sigmas=sigmas, atom_offsets=offs,
                centers=centers.astype(np.float64), boxsize=boxsize, voxelsize=vs, box=box)


def grid_origin(center, boxsize, voxelsize):
    """bb_min and nvoxels of the boxsize branch of getCenters (voxeldescriptors.py:240-243)."""
    boxsize = np.asarray(boxsize, dtype=np.float64)
    center = np.asarray(center, dtype=np.float64)
    nvox = np.ceil(boxsize / voxelsize).astype(int)
    return center - boxsize / 2, nvox
