#!/usr/bin/env python3
"""tools/tile_model.py -- what bounds the tile kernel on cfg2: a host-side census of the pair tests per voxel under the
culling schemes a wave-per-tile gather can use, on the very atoms bench.py voxelizes (tests/synth.py, config 2).

For sampled 8x8x8 tiles it counts the (atom, channel) ENTRIES
  useful      : within 5 A of the voxel                         (what the reference's O(N V) loop accepts: the floor)
  tile        : within 5 A of the tile's voxel box              (what one wave must at least look at: broadcast reads)
  x-reach     : the kernel's three buckets (all 8 planes / low 4 / high 4)       -> tests per voxel as implemented
  per-plane   : exact plane range per entry (lane-independent)                    -> what finer x buckets could reach
  quadrant    : one list per 4x4-lane quadrant, exact cull                        -> sub-wave lists (ideal, unpadded)
  quadrant+pad: the same with every (channel, class) group padded to the longest of the four lists (the four
                quadrants of a wave iterate together)
and the (channel, class) group structure (non-empty groups, entries per group).  Prints per-voxel averages and the VALU
cycle floor they imply at the measured instruction costs."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.synth import synth_config

R = 5.0


def main(ntiles=96, seed=0):
    p = synth_config(2, 1)
    xyz = p["coords"].astype(np.float64) - (p["centers"][0] - p["boxsize"] / 2)      # voxel units, voxel i at i
    sig = p["sigmas"]
    a_idx, c_idx = np.nonzero(sig)
    epos, ech, esig = xyz[a_idx], c_idx, sig[a_idx, c_idx]
    classes = {s: i for i, s in enumerate(sorted(set(esig)))}
    ecls = np.array([classes[s] for s in esig])
    rng = np.random.default_rng(seed)
    acc = {k: [] for k in ("useful", "tile", "xreach", "xreach_yz", "perplane", "quad", "quadpad", "groups", "subbuckets", "cand")}
    for _ in range(ntiles):
        t0 = rng.integers(1, 7, size=3) * 8                                           # interior tiles
        lo, hi = t0.astype(np.float64), t0 + 7.0
        near = np.all((epos > lo - 13) & (epos < hi + 13), axis=1)                    # 27 cells of 8^3 around the tile
        acc["cand"].append(near.sum())
        P, C, K = epos[near], ech[near], ecls[near]
        gap = np.maximum(np.maximum(lo - P, P - hi), 0.0)
        m = (gap ** 2).sum(1) < R * R
        P, C, K, gap = P[m], C[m], K[m], gap[m]
        n = len(P)
        acc["tile"].append(n)
        vx = np.arange(8) + t0[0]
        vy, vz = np.meshgrid(np.arange(8) + t0[1], np.arange(8) + t0[2], indexing="ij")
        dyz2 = (P[:, 1, None, None] - vy[None]) ** 2 + (P[:, 2, None, None] - vz[None]) ** 2        # [n,8,8]
        dx2 = (P[:, 0, None] - vx[None]) ** 2                                                       # [n,8]
        useful = (dx2[:, :, None, None] + dyz2[:, None] < R * R).sum() / 512.0
        acc["useful"].append(useful)
        ex = P[:, 0] - (t0[0] + 3.5)
        full = ~((0.5 - ex > R) | (ex + 0.5 > R))
        acc["xreach"].append((full * 8 + (~full) * 4).sum() / 8.0)
        gyz2 = gap[:, 1] ** 2 + gap[:, 2] ** 2
        # the kernel since the end of round 2: the reach along x shrinks with the entry's distance from the tile's y-z square
        rx2 = R * R - gyz2
        half = ((0.5 - ex > 0) & ((0.5 - ex) ** 2 > rx2)) | ((ex + 0.5 > 0) & ((ex + 0.5) ** 2 > rx2))
        acc["xreach_yz"].append(((~half) * 8 + half * 4).sum() / 8.0)
        planes = (dx2 < (R * R - gyz2)[:, None]).sum(1)
        acc["perplane"].append(planes.sum() / 8.0)
        # quadrants: lanes (y,z) in 4x4 blocks
        qlists = []
        for qy in (0, 1):
            for qz in (0, 1):
                qlo = np.array([t0[0], t0[1] + 4 * qy, t0[2] + 4 * qz], float)
                qhi = qlo + np.array([7.0, 3.0, 3.0])
                g = np.maximum(np.maximum(qlo - P, P - qhi), 0.0)
                qlists.append((g ** 2).sum(1) < R * R)
        q = np.stack(qlists)                                                          # [4,n]
        acc["quad"].append(q.sum() / 4.0)
        key = C * 16 + K
        pad = 0
        for g in np.unique(key):
            sel = key == g
            pad += q[:, sel].sum(1).max()
        acc["quadpad"].append(float(pad))
        acc["groups"].append(len(np.unique(key)))
        xr = np.where(0.5 - ex > R, 1, np.where(ex + 0.5 > R, 2, 0))
        acc["subbuckets"].append(len(np.unique(key * 3 + xr)))
    A = {k: float(np.mean(v)) for k, v in acc.items()}
    print(f"cfg2, {ntiles} interior tiles (8x8x8 voxels, 1 A), entries = (atom, channel) pairs")
    print(f"  candidates in the 27 surrounding cells      {A['cand']:8.1f}")
    print(f"  entries within 5 A of the tile box           {A['tile']:8.1f}   in {A['groups']:.1f} (channel, class) groups, {A['subbuckets']:.1f} x-reach sub-buckets")
    print(f"  pair tests per voxel")
    print(f"    useful (within 5 A of the voxel)           {A['useful']:8.1f}   = the reference's accepted pairs: the floor of any scheme")
    print(f"    every tile entry against every voxel       {A['tile']:8.1f}   efficiency {A['useful'] / A['tile']:.2f}")
    print(f"    3 x-reach buckets, reach = cutoff          {A['xreach']:8.1f}   efficiency {A['useful'] / A['xreach']:.2f}")
    print(f"    kernel today (reach from the y-z distance) {A['xreach_yz']:8.1f}   efficiency {A['useful'] / A['xreach_yz']:.2f}")
    print(f"    exact plane range per entry                {A['perplane']:8.1f}   efficiency {A['useful'] / A['perplane']:.2f}")
    print(f"    per-quadrant lists (ideal)                 {A['quad']:8.1f}   efficiency {A['useful'] / A['quad']:.2f}")
    print(f"    per-quadrant lists, groups padded          {A['quadpad']:8.1f}   efficiency {A['useful'] / A['quadpad']:.2f}")
    # VALU cycle model: per pair of entries x 8 planes: 8 v_pk_fma (4.4 cyc) + 8 v_min3 (4.2) + 5 shared (~3.4)
    cyc_per_test = (8 * 4.4 + 8 * 4.2 + 5 * 3.4) / 16.0
    tiles = 256 * 512
    simd_cycles = 1024 * 2.4e9
    for name in ("useful", "xreach", "perplane", "quadpad"):
        t = A[name] * 8 * cyc_per_test * tiles / simd_cycles * 1e3
        print(f"  pair loops alone at {cyc_per_test:.1f} VALU cycles per wave-test, {name:9s}: {t:6.3f} ms per 256-grid step"
              f"  -> {2710.7e6 / (t * 1e-3) / 8e12:5.2f} of the HBM roofline if NOTHING else ran")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 96)
