#!/usr/bin/env python3
"""tools/pair_walk_ab.py -- the pair-table walk (for_pair_run): selfdist 450 x 450 x 2 048 frames and the contact lists of 200 x 500 x 2 048,
periodic, with the chain ids of the bench legs (30 chains: a few batches per row mix wrapping and other pairs), every atom a chain of its own
(all pairs wrap) and one chain (none wraps).  MKAMD_LIB selects the build (same-box A-B; a DIAGNOSTICS build needs MKAMD_ALLOW_DIAGNOSTICS=1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib
dev = torch.device("cuda", 0)
N, F = 30000, 2048
rng = np.random.default_rng(4)
coords = torch.rand((N, 3, F), device=dev) * 66.9
box = torch.full((3, F), 66.9, device=dev)
s1 = np.sort(rng.choice(N, 200, replace=False)).astype(np.int32)
s2 = np.sort(rng.choice(N, 500, replace=False)).astype(np.int32)
ss = s2[:450].copy()
d1, d2, ds = (torch.as_tensor(x, device=dev) for x in (s1, s2, ss))
out = torch.empty((F, 450 * 449 // 2), device=dev)
ctx = _lib.default_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)


def timed(call, n=40):
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        for _ in range(8):
            call()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


row = []
only = os.environ.get("PAIR_WALK_ONLY")                                 # one of the chain settings alone (profiling passes)
for name, ch in (("30 chains", (np.arange(N) // 1000).astype(np.int32)), ("all wrap", np.arange(N, dtype=np.int32)), ("none wraps", np.zeros(N, np.int32))):
    if only and only != name:
        continue
    dch = torch.as_tensor(ch, device=dev)
    t_self = timed(lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), ds.data_ptr(), 450, ds.data_ptr(), 450, dch.data_ptr(), True, True, False, out.data_ptr()))
    t_con = timed(lambda: ctx.contacts_trajectory_dev(coords, F, box, d1, 200, d2, 500, dch, False, True, 8.0), n=20)
    t_cs = timed(lambda: ctx.contacts_trajectory_dev(coords, F, box, ds, 450, ds, 450, dch, True, True, 8.0), n=20)
    row.append(f"{name}: selfdist {t_self:.1f} us, contacts {t_con:.1f} us, selfdist contacts 450^2 {t_cs:.1f} us")
# calculate_contacts with sel1 == sel2 over a whole protein (distance.py:364): 3 000 atoms, 4.5 M pairs, 256 frames
big = torch.as_tensor(np.sort(rng.choice(N, 3000, replace=False)).astype(np.int32), device=dev)
dch = torch.as_tensor((np.arange(N) // 1000).astype(np.int32), device=dev)
c256, b256 = coords[:, :, :256].contiguous(), box[:, :256].contiguous()
t_big = timed(lambda: ctx.contacts_trajectory_dev(c256, 256, b256, big, 3000, big, 3000, dch, True, True, 4.0), n=3)
row.append(f"selfdist contacts 3 000^2 x 256 frames {t_big / 1e3:.2f} ms")
print(f"{os.path.basename(os.environ.get('MKAMD_LIB', 'libmkamd.so')):20s} " + " | ".join(row), flush=True)
