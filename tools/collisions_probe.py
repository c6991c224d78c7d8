#!/usr/bin/env python3
"""tools/collisions_probe.py -- get_collisions (distance_utils.pyx:98-121: one frame, no box) at the sizes Molecule.append(collisiondist=...) sends:
a solvated system against a molecule being inserted.  Wall clock of the whole Python call (host arrays in, list out) and of the device part."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib
from moleculekit_amd.distance_utils import get_collisions
rng = np.random.default_rng(2)
ctx = _lib.default_context(0)
for n1, n2, L in ((5000, 300, 40.0), (20000, 3000, 60.0), (60000, 3000, 90.0), (3000, 20000, 60.0)):
    c1 = rng.uniform(0, L, size=(n1, 3)).astype(np.float32)
    c2 = rng.uniform(0, L, size=(n2, 3)).astype(np.float32)
    r = get_collisions(c1, c2, 1.3)
    t0 = time.perf_counter()
    for _ in range(5):
        r = get_collisions(c1, c2, 1.3)
    dt = (time.perf_counter() - t0) / 5
    print(f"{n1:6d} x {n2:6d}: {dt * 1e3:8.2f} ms per call, {len(r) // 2} collisions, {n1 * n2 / dt / 1e9:7.2f} G pair tests/s", flush=True)
