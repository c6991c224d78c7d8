#!/usr/bin/env python3
"""tools/bench_host_dist.py -- dist_trajectory with HOST arrays in and out: 3 000 atoms x 2 048 frames, 300 x 300 pairs
(737 MB of float32 results into the caller's array, fresh each call as the reference's callers allocate it)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moleculekit_amd import distance_utils as du
rng = np.random.default_rng(2)
N, F = 3000, 2048
coords = rng.uniform(0, 40, size=(N, 3, F)).astype(np.float32); box = np.full((3, F), 40.0, np.float32)
s1 = np.arange(300, dtype=np.uint32); s2 = np.arange(1000, 1300, dtype=np.uint32); ch = (np.arange(N) // 500).astype(np.uint32)
for name, fresh in (("fresh results array", True), ("reused results array", False)):
    r = np.zeros((F, 90000), np.float32)
    if not fresh: r[:] = 1
    ts = []
    for _ in range(4):
        if fresh: r = np.zeros((F, 90000), np.float32)
        t0 = time.perf_counter(); du.dist_trajectory(coords, box, s1, s2, ch, False, True, r); ts.append(time.perf_counter() - t0)
    print(f"{name}: {min(ts) * 1e3:.1f} ms per call ({r.nbytes / 1e6:.0f} MB of results)")
