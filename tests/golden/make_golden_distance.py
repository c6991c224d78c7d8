#!/usr/bin/env python3
"""Golden fixtures for the distance_utils row, generated from the REAL reference's
compiled moleculekit.distance_utils (see make_golden.py for how the reference is built in a scratch dir):

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_distance.py

Only data is stored: seeded synthetic inputs and the reference's outputs.
"""
import os
import sys

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)
from moleculekit import distance_utils as du  # noqa: E402
import json  # noqa: E402


def main():
    rng = np.random.default_rng(81)
    N, F = 60, 7
    coords = rng.uniform(-25, 25, size=(N, 3, F)).astype(np.float32)
    box = rng.uniform(18, 30, size=(3, F)).astype(np.float32)
    chains = rng.integers(0, 4, size=N).astype(np.uint32)
    masses = rng.uniform(1, 32, size=N).astype(np.float32)
    out = dict(coords=coords, box=box, chains=chains, masses=masses)

    # ---- dist_trajectory: cross / self selections, with and without pbc ----
    sel1 = np.sort(rng.choice(N, 9, replace=False)).astype(np.uint32)
    sel2 = np.sort(rng.choice(N, 13, replace=False)).astype(np.uint32)
    out["sel1"], out["sel2"] = sel1, sel2
    for pbc in (0, 1):
        r = np.zeros((F, len(sel1) * len(sel2)), np.float32)
        du.dist_trajectory(coords, box, sel1, sel2, chains, False, bool(pbc), r)
        out[f"dist_cross_pbc{pbc}"] = r
        n = len(sel2)
        r = np.zeros((F, n * (n - 1) // 2), np.float32)
        du.dist_trajectory(coords, box, sel2, sel2, chains, True, bool(pbc), r)
        out[f"dist_self_pbc{pbc}"] = r
    # zero box with pbc=False (what the reference's callers pass for periodic=None, projections/util.py:30-37)
    r = np.zeros((F, len(sel1) * len(sel2)), np.float32)
    du.dist_trajectory(coords, np.zeros_like(box), sel1, sel2, np.zeros(N, np.uint32), False, False, r)
    out["dist_cross_zero_box"] = r

    # ---- contacts_trajectory ----
    res = du.contacts_trajectory(coords, box, sel1, sel2, chains, False, True, 12.0)
    out["contacts_counts"] = np.array([len(x) // 2 for x in res], np.int64)
    out["contacts_flat"] = np.concatenate([np.asarray(x, np.int64) for x in res]) if sum(len(x) for x in res) else np.zeros(0, np.int64)
    res = du.contacts_trajectory(coords, box, sel2, sel2, chains, True, False, 15.0)
    out["contacts_self_counts"] = np.array([len(x) // 2 for x in res], np.int64)
    out["contacts_self_flat"] = np.concatenate([np.asarray(x, np.int64) for x in res])

    # ---- get_collisions ----
    c1 = np.ascontiguousarray(coords[:20, :, 0]); c2 = np.ascontiguousarray(coords[20:50, :, 0])
    out["collisions"] = np.asarray(du.get_collisions(c1, c2, 14.0), np.int64)

    # ---- reductions: closest / com, all-vs-all, self and pairs ----
    perm = rng.permutation(N)
    groups1 = [sorted(perm[i * 3:(i + 1) * 3].tolist()) for i in range(5)]          # 5 groups of 3
    groups2 = [sorted(perm[20 + i * 4:20 + (i + 1) * 4].tolist()) for i in range(6)]  # 6 groups of 4
    out["groups1"] = np.array(groups1, np.int64); out["groups2"] = np.array(groups2, np.int64)
    ch1 = np.array([chains[g[0]] for g in groups1], np.uint32)
    ch2 = np.array([chains[g[0]] for g in groups2], np.uint32)
    out["gchains1"], out["gchains2"] = ch1, ch2
    for r1 in (0, 1):
        for r2 in (0, 1):
            for pbc in (0, 1):
                r = np.zeros((F, 30), np.float32)
                du.dist_trajectory_reduction(coords, box, groups1, groups2, ch1, ch2, False, bool(pbc), masses, r1, r2, r)
                out[f"red_{r1}{r2}_pbc{pbc}"] = r
    r = np.zeros((F, 15), np.float32)
    du.dist_trajectory_reduction(coords, box, groups2, groups2, ch2, ch2, True, True, masses, 0, 0, r)
    out["red_self"] = r
    r = np.zeros((F, 5), np.float32)
    du.dist_trajectory_reduction_pairs(coords, box, groups1, groups2[:5], ch1, ch2[:5], True, masses, 0, 1, r)
    out["red_pairs"] = r

    # ---- cdist / pdist / squareform (incl. dimensions other than 3) ----
    for D in (1, 2, 3, 5):
        a = rng.normal(size=(11, D)).astype(np.float32) * 7
        b = rng.normal(size=(17, D)).astype(np.float32) * 7
        r = np.zeros((11, 17), np.float32)
        du.cdist(a, b, r)
        out[f"cdist_a{D}"], out[f"cdist_b{D}"], out[f"cdist_r{D}"] = a, b, r
        r = np.zeros(17 * 16 // 2, np.float32)
        du.pdist(b, r)
        out[f"pdist_r{D}"] = r
    out["squareform"] = np.array(du.squareform(out["pdist_r3"]))
    np.savez_compressed(os.path.join(OUT, "distance_cases.npz"), **out)
    print("wrote distance_cases.npz", os.path.getsize(os.path.join(OUT, "distance_cases.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
