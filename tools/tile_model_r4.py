#!/usr/bin/env python3
"""tools/tile_model_r4.py -- round 4: pair-loop TRIPS per tile (not only tests per voxel) of the tile kernel on cfg2 under
the list layouts that keep ONE copy of every entry in LDS.

Unit: a "plane-op" = one packed fma + one min3 (two entries against one x-plane); a full trip = 5 shared packed
operations + 8 plane-ops, a half-plane trip 5 + 4.  The kernel today runs, per (channel, class, x-reach) sub-bucket,
floor(n / 2) trips and a single-entry tail for an odd n.

Schemes (all exact: a lane only ever tests a SUPERSET of the entries that can reach its voxels)
  today        : three x-reach sub-buckets per group, broadcast entries
  ywin2        : every sub-bucket sorted [only the low-y half | both | only the high-y half]; lanes 0-31 walk the window
                 [start, start + 2T), lanes 32-63 the window [end - 2T, end), T = the longer of the two (per-lane LDS address)
  ywin4        : four row pairs, seven zones
  ywin2 + zpair: additionally ... (not modelled)
Prints trips, instruction estimates and the group census the direct evaluation of small groups needs."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.synth import synth_config

R = 5.0
R2 = R * R


def reach_zone(e, g_other2, lo_edge, hi_edge):
    """0 = reaches both halves, 1 = only the low half (coordinates < 0), 2 = only the high half; e relative to the tile
    centre, the halves' nearest rows / planes at -0.5 and +0.5 (x_reach of kernels.h with the roles of the axes swapped)"""
    r2 = R2 - g_other2
    lo = 0.5 - e          # distance to the nearest row of the upper half
    hi = e + 0.5
    only_low = (lo > 0) & (lo * lo > r2)
    only_high = (hi > 0) & (hi * hi > r2)
    return np.where(only_low, 1, np.where(only_high, 2, 0))


def main(ntiles=200, seed=0):
    p = synth_config(2, 1)
    xyz = p["coords"].astype(np.float64) - (p["centers"][0] - p["boxsize"] / 2)
    sig = p["sigmas"]
    a_idx, c_idx = np.nonzero(sig)
    epos, ech, esig = xyz[a_idx], c_idx, sig[a_idx, c_idx]
    classes = {s: i for i, s in enumerate(sorted(set(esig)))}
    ecls = np.array([classes[s] for s in esig])
    rng = np.random.default_rng(seed)
    keys = ("entries", "groups", "g1", "g2", "g1_planeops", "sub_today", "odd_today", "full_today", "half_today",
            "tail_full", "tail_half", "full_w2", "half_w2", "pads_w2", "sub_w2", "full_w4", "half_w4",
            "tests_today", "tests_w2", "tests_w4", "full_w2z", "half_w2z")
    acc = {k: [] for k in keys}
    for _ in range(ntiles):
        t0 = rng.integers(1, 7, size=3) * 8
        lo, hi = t0.astype(np.float64), t0 + 7.0
        gap = np.maximum(np.maximum(lo - epos, epos - hi), 0.0)
        m = (gap ** 2).sum(1) < R2
        P, C, K, gap = epos[m] - (t0 + 3.5), ech[m], ecls[m], gap[m]
        n = len(P)
        acc["entries"].append(n)
        xr = reach_zone(P[:, 0], gap[:, 1] ** 2 + gap[:, 2] ** 2, 0, 0)
        yz = reach_zone(P[:, 1], gap[:, 0] ** 2 + gap[:, 2] ** 2, 0, 0)
        zz = reach_zone(P[:, 2], gap[:, 0] ** 2 + gap[:, 1] ** 2, 0, 0)
        # four row pairs (rows -3.5,-2.5 | -1.5,-0.5 | 0.5,1.5 | 2.5,3.5): the row-pair interval an entry reaches
        ry = np.sqrt(np.maximum(R2 - gap[:, 0] ** 2 - gap[:, 2] ** 2, 0.0))
        rows = np.arange(8) - 3.5
        reach_rows = np.abs(rows[None, :] - P[:, 1:2]) < ry[:, None]                  # [n,8]
        rp = reach_rows.reshape(n, 4, 2).any(2)                                       # [n,4] row pairs reached
        first = np.argmax(rp, 1)
        last = 3 - np.argmax(rp[:, ::-1], 1)
        grp = C * 16 + K
        st = {k: 0 for k in keys}
        ug = np.unique(grp)
        st["groups"] = len(ug)
        for g in ug:
            sel = grp == g
            ng = sel.sum()
            if ng == 1:
                st["g1"] += 1
                st["g1_planeops"] += 8 if xr[sel][0] == 0 else 4
            if ng == 2:
                st["g2"] += 1
            for x in (0, 1, 2):
                s2 = sel & (xr == x)
                nn = int(s2.sum())
                if nn == 0:
                    continue
                planes = 8 if x == 0 else 4
                st["sub_today"] += 1
                st["odd_today"] += nn & 1
                trips = nn // 2
                st["full_today" if x == 0 else "half_today"] += trips
                if nn & 1:
                    st["tail_full" if x == 0 else "tail_half"] += 1
                st["tests_today"] += nn * planes
                # --- y windows, two halves: [L | M | H] with a sentinel in the pad slot
                nL, nM, nH = int((s2 & (yz == 1)).sum()), int((s2 & (yz == 0)).sum()), int((s2 & (yz == 2)).sum())
                npad = (nn + 1) & ~1
                t_low = (nL + nM + 1) // 2
                t_high = (npad - (nL & ~1)) // 2
                T = max(t_low, t_high)
                st["full_w2" if x == 0 else "half_w2"] += T
                st["pads_w2"] += nn & 1
                st["sub_w2"] += 1
                st["tests_w2"] += 2 * T * planes
                # --- additionally the z halves pair off: (only-low-z, only-high-z) entries of a y zone share a slot pair ... not modelled
                # --- four row pairs: entries sorted by the FIRST row pair they reach then by the last (7 zones in practice);
                #     window of row pair r = every entry with first <= r <= last, as a contiguous range of the order
                f, l = first[s2], last[s2]
                order = np.lexsort((l, f))
                f, l = f[order], l[order]
                Tm = 0
                for r in range(4):
                    need = np.nonzero((f <= r) & (l >= r))[0]
                    if len(need):
                        a, b = need[0] & ~1, need[-1] + 1
                        Tm = max(Tm, (b - a + 1) // 2)
                st["full_w4" if x == 0 else "half_w4"] += Tm
                st["tests_w4"] += 2 * Tm * planes
        for k in keys:
            if k != "entries":
                acc[k].append(st[k])
    A = {k: float(np.mean(v)) for k, v in acc.items()}
    FULL, HALF, TAILF, TAILH = 22, 14, 21, 13
    print(f"cfg2, {ntiles} interior tiles; entries within reach of the tile {A['entries']:.1f} in {A['groups']:.1f} groups "
          f"({A['g1']:.1f} with one entry, {A['g2']:.1f} with two)")
    it = A["full_today"] * FULL + A["half_today"] * HALF + A["tail_full"] * TAILF + A["tail_half"] * TAILH
    print(f"today : {A['sub_today']:.1f} sub-buckets ({A['odd_today']:.1f} odd), {A['full_today']:.1f} full + {A['half_today']:.1f} half trips, "
          f"{A['tail_full']:.1f} + {A['tail_half']:.1f} tails -> {it:.0f} VALU instructions in loops, {A['tests_today'] / 8:.1f} tests per voxel")
    iw = A["full_w2"] * FULL + A["half_w2"] * HALF
    print(f"ywin2 : {A['full_w2']:.1f} full + {A['half_w2']:.1f} half trips, no tails ({A['pads_w2']:.1f} sentinel pads) -> {iw:.0f} "
          f"({100 * (iw / it - 1):+.1f} %), {A['tests_w2'] / 8:.1f} tests per voxel")
    i4 = A["full_w4"] * FULL + A["half_w4"] * HALF
    print(f"ywin4 : {A['full_w4']:.1f} full + {A['half_w4']:.1f} half trips -> {i4:.0f} ({100 * (i4 / it - 1):+.1f} %), {A['tests_w4'] / 8:.1f} tests per voxel")
    print(f"single-entry groups: {A['g1']:.1f} per tile, {A['g1_planeops']:.1f} plane evaluations")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
