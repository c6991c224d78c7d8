"""tools/dist_few_frames_probe.py -- dist_trajectory on ONE structure or a handful of frames (MetricDistance on a PDB, a docking pose set): which kernel
the library takes and what the call costs, rectangular and triangular shapes; microseconds per call and G distances/s."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculekit_amd import _lib
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
lib = _lib.load()
rng = np.random.default_rng(4)
N = 30000
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
SELF_ONLY = os.environ.get("PROBE_SELF") == "1"        # selfdist alone, both kernels (the triangular row kernel / the pair-table kernel: avoid bit 64)
for F in (1, 4, 16, 64) if not SELF_ONLY else (1, 8, 16, 32, 64, 512):
    coords = torch.rand((N, 3, F), device=dev) * 66.9
    box = torch.full((3, F), 66.9, device=dev)
    chains = torch.as_tensor((np.arange(N) // 1000).astype(np.int32), device=dev)
    for n1, n2, selfd in ((300, 30, False), (3000, 300, False), (5000, 5000, False), (1000, 1000, True), (5000, 5000, True)) if not SELF_ONLY else \
            ((250, 250, True), (450, 450, True), (1000, 1000, True), (2000, 2000, True), (5000, 5000, True)):
        s2 = np.sort(rng.choice(N, n2, replace=False)).astype(np.int32)
        s1 = s2 if selfd else np.sort(rng.choice(N, n1, replace=False)).astype(np.int32)
        d1, d2 = torch.as_tensor(s1, device=dev), torch.as_tensor(s2, device=dev)
        P = int(lib.mkamd_dist_count_pairs(n1, n2, int(selfd)))
        out = torch.empty((F, P), device=dev)
        row = []
        if SELF_ONLY and F * P * 4 > 6e9:
            continue
        for pbc, mask in ((False, 0), (True, 0)) if not SELF_ONLY else ((False, 16), (False, 64), (True, 16), (True, 64)):
            ctx.set_dist_kernels(mask)
            ms = t(lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2, chains.data_ptr(), selfd, pbc, False, out.data_ptr()))
            row.append(f"{'pbc' if pbc else 'open'} {ms * 1e3:8.1f} us {F * P / ms / 1e6:7.1f} G/s [{ctx.last_dist_kernel()}]")
        ctx.set_dist_kernels(0)
        print(f"F={F:3d} {n1:5d} x {n2:5d} {'self' if selfd else 'rect'} ({F * P * 4 / 1e6:7.1f} MB): " + " | ".join(row), flush=True)
