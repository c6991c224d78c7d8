#!/usr/bin/env python3
"""tools/dist_rows_ab.py -- dist_trajectory 200 x 500 x 2 048 frames, periodic, with the chain ids of the bench leg (mixed among a wave's second
atoms) and with the reference's periodic="selections" ids (1 / 2: every pair wraps).  MKAMD_LIB selects the build (same-box A-B)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib
dev = torch.device("cuda", 0)
N, F, n1, n2 = 30000, 2048, 200, 500
rng = np.random.default_rng(4)
coords = torch.rand((N, 3, F), device=dev) * 66.9
box = torch.full((3, F), 66.9, device=dev)
s1 = np.sort(rng.choice(N, n1, replace=False)).astype(np.int32)
s2 = np.sort(rng.choice(N, n2, replace=False)).astype(np.int32)
d1, d2 = torch.as_tensor(s1, device=dev), torch.as_tensor(s2, device=dev)
out = torch.empty((F, n1 * n2), device=dev)
ctx = _lib.default_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
sel_ids = np.ones(N, np.int32); sel_ids[s2] = 2
row = []
for name, ch in (("chains (mixed)", (np.arange(N) // 1000).astype(np.int32)), ("selections (all wrap)", sel_ids), ("one chain (none wraps)", np.zeros(N, np.int32))):
    dch = torch.as_tensor(ch, device=dev)
    call = lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2, dch.data_ptr(), False, True, False, out.data_ptr())
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        for _ in range(16):
            call()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        call()
    e1.record(); torch.cuda.synchronize()
    row.append(f"{name}: {e0.elapsed_time(e1) / 40 * 1e3:.1f} us")
print(f"{os.path.basename(os.environ.get('MKAMD_LIB', 'libmkamd.so')):22s} " + " | ".join(row), flush=True)
