#!/usr/bin/env python3
"""Golden fixture for BASELINE.json configs[3] (cfg4) at its REAL shape: one frame of 30 000 atoms in a 66.9 A
periodic box on a 48^3 grid @ 1 A, from the REAL reference.

The reference's voxelizer has no periodic mode; the periodic semantic is the 27-image
composition of its own kernel: calculate_occupancy called with centres shifted by k*box, k in {-1,0,1}^3,
max-accumulating into the SAME results buffer (legal: occupancy_utils.pyx:61 max-accumulates in place).  27 images
suffice here: the atoms are wrapped and the grid lies inside the box.

Run in the build container only (see make_golden.py for the one-off build of the reference):

    MOLECULEKIT_REF_BUILD=/tmp/mkbuild python3 tests/golden/make_golden_cfg4.py

Stores DATA only: sampled voxel indices, the reference's values there, per-channel sums / maxima / non-zero counts
of the full grid.  The inputs are regenerated from the seeded generator (tests/synth.py) by the test.
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

REF_BUILD = os.environ.get("MOLECULEKIT_REF_BUILD", "/tmp/mkbuild")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF_BUILD)
sys.path.insert(0, os.path.join(OUT, "..", ".."))


def _chunk(args):
    from moleculekit.occupancy_utils import calculate_occupancy
    centers, coords, sigmas, box = args
    res = np.zeros((centers.shape[0], sigmas.shape[1]))
    for kx in (-1, 0, 1):
        for ky in (-1, 0, 1):
            for kz in (-1, 0, 1):
                shifted = np.ascontiguousarray(centers - np.array([kx, ky, kz], np.float64) * box)
                calculate_occupancy(shifted, coords, sigmas, res)
    return res


def main():
    from moleculekit.tools.voxeldescriptors import getCenters
    from tests.synth import synth_config
    p = synth_config(4, 2)
    n = int(p["atom_offsets"][1])
    out = {}
    for frame in (0, 1):
        s, e = frame * n, (frame + 1) * n
        coords = np.ascontiguousarray(p["coords"][s:e], np.float32)
        sigmas = np.ascontiguousarray(p["sigmas"][s:e], np.float64)
        box = p["box"][frame].astype(np.float64)
        centers, nv = getCenters(None, boxsize=list(p["boxsize"]), center=list(p["centers"][frame]), voxelsize=p["voxelsize"])
        chunks = np.array_split(np.arange(centers.shape[0]), 96)
        with Pool(min(os.cpu_count() or 1, 48)) as pool:
            parts = pool.map(_chunk, [(centers[c], coords, sigmas, box) for c in chunks])
        f = np.concatenate(parts)
        rng = np.random.default_rng(404 + frame)
        samp = np.sort(rng.choice(f.shape[0], 16384, replace=False))
        out.update({f"f{frame}_sample_idx": samp, f"f{frame}_sample_features": f[samp], f"f{frame}_channel_sums": f.sum(0),
                    f"f{frame}_nonzero": np.count_nonzero(f, axis=0), f"f{frame}_channel_max": f.max(0)})
        out["nvoxels"] = nv
        print("frame", frame, "done; channel sums", f.sum(0))
    np.savez_compressed(os.path.join(OUT, "cfg4_full_sampled.npz"), **out)


if __name__ == "__main__":
    main()
