"""One-off wider sweep of contacts_trajectory / get_collisions on the GPU (not collected by pytest): random atom counts, frame counts
(1 .. 200: the few-frame kernel with lanes along the second atoms, the rectangular kernels with lanes along frames, several slabs),
selection shapes (rows that end inside a run of 16 / a tile of 64, rows too short for the rectangular kernels: the pair-table walk),
selfdist with equal and unequal selections, boxes down to a few Angstrom (many image shifts), repeated and unsorted atoms, chain ids
that make every / some / no pair wrap.  Every list against the oracle's squared distances in the reference's (frame, i, j) order.
    python tests/sweep_gpu_contacts.py [first_seed] [count]"""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from moleculekit_amd.distance_utils import contacts_trajectory, get_collisions

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad, seen = 0, collections.Counter()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(2, 900))
    F = int(rng.choice([1, 2, 7, 16, 17, 63, 64, 65, 130, 200]))
    L = float(rng.choice([3.0, 11.0, 40.0]))
    c = rng.uniform(-1.5 * L, 1.5 * L, size=(N, 3, F)).astype(np.float32)
    b = (L * rng.uniform(0.8, 1.2, size=(3, F))).astype(np.float32)
    mode = rng.integers(0, 3)
    ch = np.zeros(N, np.uint32) if mode == 0 else rng.integers(0, int(rng.integers(2, 6)), size=N).astype(np.uint32) if mode == 1 else np.arange(N, dtype=np.uint32)
    selfdist = bool(rng.random() < 0.35)
    n2 = int(rng.choice([1, 5, 23, 24, 52, 63, 64, 65, 100, 128, 200, 260, 500]))
    n1 = int(rng.choice([1, 7, 16, 17, 40, 100, 200]))
    if selfdist and rng.random() < 0.6:
        n1 = n2
    n1, n2 = min(n1, 4 * N), min(n2, 4 * N)
    s2 = rng.integers(0, N, size=n2).astype(np.uint32)
    s1 = s2[:n1].copy() if (selfdist and n1 <= n2 and rng.random() < 0.7) else rng.integers(0, N, size=n1).astype(np.uint32)
    pbc = bool(rng.random() < 0.6)
    thr = float(rng.choice([0.3, 0.5, 0.8]) * L)
    d2 = oracle.dist_trajectory(c, b, s1, s2, ch, selfdist, pbc, squared=True)
    pairs = [(i, j) for i in range(n1) for j in (range(i + 1, n2) if selfdist else range(n2))]
    thr2 = np.float32(thr) * np.float32(thr)
    want = []
    with np.errstate(all="ignore"):
        for f in range(F):
            want.append([int(v) for k in np.nonzero(d2[f] <= thr2)[0] for v in (s1[pairs[k][0]], s2[pairs[k][1]])])
    got = contacts_trajectory(c, b, s1, s2, ch, selfdist, pbc, thr)
    seen[("selfdist" if selfdist else "rect", "pbc" if pbc else "open", "few" if F <= 16 else "frames")] += 1
    if got != want:
        bad += 1
        print("FAIL seed", seed, dict(N=N, F=F, n1=n1, n2=n2, selfdist=selfdist, pbc=pbc, L=L, thr=thr), flush=True)
    if seed % 5 == 0:                                            # get_collisions: one frame, no box, row indices
        a, bb = c[s1.astype(np.int64), :, 0].copy(), c[s2.astype(np.int64), :, 0].copy()
        diff = a[:, None, :] - bb[None, :, :]
        e = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        wantc = np.argwhere(e <= thr2).astype(np.int64).ravel().tolist()
        seen[("get_collisions",)] += 1
        if get_collisions(a, bb, thr) != wantc:
            bad += 1
            print("FAIL get_collisions seed", seed, dict(n1=n1, n2=n2, thr=thr), flush=True)
print("seeds", first, "..", first + count - 1, ":", bad, "calls differ from the reference's lists;", dict(seen))
