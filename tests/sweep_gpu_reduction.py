#!/usr/bin/env python3
"""tests/sweep_gpu_reduction.py [first_seed] [count] -- random dist_trajectory_reduction[_pairs] calls on the GPU against the oracle, bit for bit
(the evidence session runs it on the final build; every seed draws its own atom count, frame count (two in five: 1-17 frames), ragged groups of 1-24 atoms, chain ids,
reductions closest / com, self / pairs modes, periodic or open, boxes that put some separations near half a box length)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moleculekit_amd import distance_utils as du
from oracle import oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    N, F = int(rng.integers(5, 400)), int(rng.integers(1, 150))
    if rng.random() < 0.4:
        F = int(rng.integers(1, 18))                              # calls of few frames: k_dist_reduction_few (lanes along the second groups)
    L = float(rng.uniform(12, 60))
    coords = rng.uniform(-0.7 * L, 0.7 * L, size=(N, 3, F)).astype(np.float32)
    box = (L * rng.uniform(0.9, 1.1, size=(3, F))).astype(np.float32)
    masses = rng.uniform(1, 40, N).astype(np.float32)
    hi = int(rng.choice([1, 4, 9, 24]))
    ng1, ng2 = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    if rng.random() < 0.15:
        ng2 = int(rng.integers(200, 600))                         # several blocks along a row of second groups
    mk = lambda n: [rng.choice(N, int(rng.integers(1, min(hi, N) + 1)), replace=False).tolist() for _ in range(n)]
    mode = int(rng.integers(3))                                   # 0 all-vs-all, 1 self, 2 pairs
    g1 = mk(ng1)
    g2 = g1 if mode == 1 else mk(ng1 if mode == 2 else ng2)
    ch1 = rng.integers(0, 3, len(g1)).astype(np.uint32)
    ch2 = ch1 if mode == 1 else rng.integers(0, 3, len(g2)).astype(np.uint32)
    pbc = bool(rng.integers(2))
    r1, r2 = (0, 0) if rng.random() < 0.6 else (int(rng.integers(2)), int(rng.integers(2)))
    want = oracle.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, mode == 1, pbc, masses, r1, r2, pairs=mode == 2)
    got = np.zeros_like(want)
    if mode == 2:
        du.dist_trajectory_reduction_pairs(coords, box, g1, g2, ch1, ch2, pbc, masses, r1, r2, got)
    else:
        du.dist_trajectory_reduction(coords, box, g1, g2, ch1, ch2, mode == 1, pbc, masses, r1, r2, got)
    if not np.array_equal(got, want, equal_nan=True):
        bad += 1
        print(f"seed {seed}: MISMATCH (N {N}, F {F}, groups {len(g1)} x {len(g2)}, mode {mode}, pbc {pbc}, r {r1}{r2}): "
              f"{int((got != want).sum())} of {want.size} values differ, worst {np.nanmax(np.abs(got - want)):.3g}", flush=True)
print(f"dist_trajectory_reduction sweep, seeds {first}..{first + count - 1}: {count - bad} of {count} calls bit-exact with the oracle", flush=True)
sys.exit(1 if bad else 0)
