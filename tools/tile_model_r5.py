#!/usr/bin/env python3
"""tools/tile_model_r5.py -- round 5 ("Next round" item 1): would a COLUMN WALK pay?  A wave (or a small workgroup) walks the
8 x-adjacent tiles of one (y, z) column of a 64^3 grid and culls the candidate cells against the column's y-z square ONCE, instead of
every tile culling its 27 cells on its own.  The model prices every phase with the MEASURED instruction counts of the shipped kernel
(profiles/r5_tile_instruction_map.txt: PMC SQ_INSTS_VALU of builds with parts compiled out) and the census of the cfg2 workload
(candidates, y-z survivors, survivors, entries per tile; tools/tile_model_r4.py has the pair-trip census).

Gate (VERDICT): build it only if the model shows <= 7 700 VALU per tile (-10 % of 8 533)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.synth import synth_config

R, R2 = 5.0, 25.0
MEASURED = dict(total=8533, pair=4955, flush=865, cull=919, hist=427, place=471, class_misc=25, epilogue=496, rest=375)


def census(ntiles=300, seed=0):
    p = synth_config(2, 1)
    xyz = p["coords"].astype(np.float64) - (p["centers"][0] - p["boxsize"] / 2)      # voxel units, voxel 0 at 0
    sig = p["sigmas"]
    has = (sig != 0).any(1)
    pos = xyz[has]
    nent = (sig[has] != 0).sum(1)
    rng = np.random.default_rng(seed)
    acc = dict(cand=[], cand_new_layer=[], yz=[], yz3=[], surv=[], entries=[])
    cell = np.floor((pos + 0.5) / 8.0).astype(int)                                   # a cell holds [8c - 0.5, 8c + 7.5)
    for _ in range(ntiles):
        t = rng.integers(1, 7, size=3)
        t0 = t * 8
        lo, hi = t0.astype(float), t0 + 7.0
        in27 = np.all(np.abs(cell - t) <= 1, axis=1)
        # the z-run trimming of find_candidate_runs is ignored (cells are tile-sized here: nothing to trim)
        acc["cand"].append(in27.sum())
        gap = np.maximum(np.maximum(lo - pos, pos - hi), 0.0)
        yz_ok = in27 & (gap[:, 1] ** 2 + gap[:, 2] ** 2 < R2)                        # the column's y-z cull: the same for every tile of the column
        acc["yz3"].append(yz_ok.sum())                                               # ... of the 3 x-layers a tile looks at
        acc["yz"].append((yz_ok & (cell[:, 0] == t[0])).sum())                       # ... of one x-layer (what a new tile adds)
        acc["cand_new_layer"].append((in27 & (cell[:, 0] == t[0])).sum())
        s = yz_ok & ((gap ** 2).sum(1) < R2)
        acc["surv"].append(s.sum())
        acc["entries"].append(nent[s].sum())
    return {k: float(np.mean(v)) for k, v in acc.items()}


def main():
    c = census()
    M = MEASURED
    ch = lambda n: n / 64.0
    print(f"cfg2 census per interior tile: candidates {c['cand']:.0f} ({ch(c['cand']):.1f} chunks of 64), of one x-layer {c['cand_new_layer']:.0f}; "
          f"pass the column's y-z cull: {c['yz3']:.0f} of the tile's three layers, {c['yz']:.0f} per layer; survive the tile's cull {c['surv']:.0f} "
          f"({c['entries']:.0f} entries)")
    cull_per_chunk = M["cull"] / np.ceil(ch(c["cand"]))
    hist_per_chunk = M["hist"] / np.ceil(ch(c["surv"]))
    place_per_chunk = M["place"] / np.ceil(ch(c["surv"]))
    print(f"measured per chunk of 64 records: cull {cull_per_chunk:.1f}, histogram {hist_per_chunk:.1f}, placement {place_per_chunk:.1f} VALU "
          f"(the survivor passes gather a record again, decode it and run the per-channel loop)")
    today = M["cull"] + M["hist"] + M["place"] + M["rest"]
    print(f"today: cull {M['cull']} + histogram {M['hist']} + placement {M['place']} + prologue / runs / scan / stores {M['rest']} = {today} of {M['total']} per tile")
    # ---- (A) one wave walks the column, the y-z survivors of the three live layers as 16-bit codes in registers (12 VGPRs) ----
    # per tile: the NEW layer's candidates are culled in y-z only (the x test is gone: ~5 of 38 instructions), every live layer's
    # codes are walked twice (histogram, placement) -- the tile's own cull now happens inside those walks, so they see the y-z
    # survivors of three layers, not the tile's survivors
    cullA = np.ceil(ch(c["cand_new_layer"])) * (cull_per_chunk - 5.0)
    walkA = np.ceil(ch(c["yz"])) * 3.0                       # chunks per survivor pass (per-layer code registers, ragged)
    histA = walkA * (hist_per_chunk + 6.0)                   # + the tile cull and its ballot inside the walk
    placeA = walkA * (place_per_chunk + 6.0)
    restA = M["rest"] - 0.875 * 120.0                        # prologue + candidate runs (~120 of the 375) once per column of 8 tiles
    totalA = M["total"] - today + cullA + histA + placeA + restA
    print(f"(A) column walk, codes in registers: cull {cullA:.0f} + histogram {histA:.0f} + placement {placeA:.0f} + rest {restA:.0f} "
          f"-> {totalA:.0f} VALU per tile ({100 * (totalA / M['total'] - 1):+.1f} %)   [gate: <= 7 700]")
    # ---- (B) a 4-wave workgroup per column, the y-z survivors of the live layers DECODED in LDS (x, y, z, ids: 16 B each) ----
    # the survivor passes read LDS instead of gathering records (~ -14 of ~55 instructions per chunk), but six live layers
    # (tiles tx .. tx+3 in flight) are 6 x 240 x 16 B = 23 KB per workgroup on top of 4 x 9.3 KB: 60 KB -> 2 workgroups per CU =
    # 2 waves per SIMD instead of 4.  Round 1's occupancy curve of this kernel (1.25 / 2 / 2.75 / 3.25 waves per SIMD: 0.69 / 0.48 /
    # 0.43 / 0.40 ms): 2 waves per SIMD cost +20 % of the kernel's time.
    histB = walkA * (hist_per_chunk + 6.0 - 14.0)
    placeB = walkA * (place_per_chunk + 6.0 - 14.0)
    totalB = M["total"] - today + cullA / 1.0 + histB + placeB + restA
    print(f"(B) workgroup per column, decoded survivors in LDS: {totalB:.0f} VALU per tile ({100 * (totalB / M['total'] - 1):+.1f} %) "
          f"at HALF the occupancy (+20 % time by the round-1 curve): net slower")
    # ---- (C) the sort once per column (k_voxelize_items' scheme): without a per-tile cull the pair loops see the y-z survivors ----
    more = c["yz3"] / c["surv"]
    print(f"(C) entries sorted once per column, tiles take whole x-layers: the pair loops would test {c['yz3']:.0f} records instead of "
          f"{c['surv']:.0f} (x {more:.2f}) and lose the x-reach sub-buckets: pair loops {M['pair']} -> >= {M['pair'] * more:.0f}: far slower")
    print("verdict: no variant comes near the -10 % gate -- the cull pass a column walk removes (919 VALU per tile) is cheaper than what its\n"
          "survivor passes gain by walking the y-z survivors of three layers (462 records) instead of the tile's own survivors (286): not built.")


if __name__ == "__main__":
    main()
