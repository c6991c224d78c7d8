#!/usr/bin/env python3
"""tools/bench_contacts.py -- wall-clock of contacts_trajectory (host arrays in, contact list out: copies included) on a
synthetic trajectory: 3 000 atoms x 512 frames, 300 x 600 pairs, periodic, 5 A threshold.  MKAMD_LIB selects the build."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moleculekit_amd import distance_utils as du

rng = np.random.default_rng(3)
N, F = 3000, 512
L = 31.0
c0 = rng.uniform(0, L, size=(N, 3)).astype(np.float32)
coords = np.ascontiguousarray((c0[:, :, None] + rng.normal(0, 0.3, size=(N, 3, F))).astype(np.float32))
box = np.full((3, F), L, dtype=np.float32)
sel1 = np.arange(0, 300, dtype=np.uint32); sel2 = np.arange(1000, 1600, dtype=np.uint32)
chains = (np.arange(N) // 500).astype(np.uint32)
res = du.contacts_trajectory(coords, box, sel1, sel2, chains, False, True, 5.0)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); res = du.contacts_trajectory(coords, box, sel1, sel2, chains, False, True, 5.0); ts.append(time.perf_counter() - t0)
n = sum(len(r) for r in res) if isinstance(res, (list, tuple)) else len(res)
print(f"contacts_trajectory: {min(ts) * 1e3:.2f} ms per call (best of 5), {180000 * F / min(ts) / 1e9:.1f} G pair-frames/s, {n} list entries")
