"""tools/dist_shapes_probe.py -- dist_trajectory at the shapes MetricDistance / MetricSelfDistance usually call it with (protein
CA x ligand atoms, both ways round; all CA pairs), 30 000 atoms x 2 048 frames: ms per call and GB/s of result."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculekit_amd import _lib
dev = torch.device("cuda", 0)
N, F = 30000, 2048
rng = np.random.default_rng(4)
coords = torch.rand((N, 3, F), device=dev) * 66.9
box = torch.full((3, F), 66.9, device=dev)
chains = torch.as_tensor((np.arange(N) // 1000).astype(np.int32), device=dev)
ctx = _lib.default_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
lib = _lib.load()
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
SHAPES_ODD = ((450, 450, True), (300, 300, True), (1000, 1000, True), (17, 130, False)) if os.environ.get("PROBE_ODD") else None
if os.environ.get("PROBE_SHAPES"):                                    # "300x60,30x300": rectangular shapes of one's own
    SHAPES_ODD = tuple(tuple(int(v) for v in sh.split("x")) + (False,) for sh in os.environ["PROBE_SHAPES"].split(","))
for n1, n2, selfd in SHAPES_ODD or ((300, 30, False), (30, 300, False), (300, 60, False), (1000, 30, False), (300, 300, True), (450, 450, True), (1000, 1000, True), (300, 300, False),
                      (200, 500, False)):
    s2 = np.sort(rng.choice(N, n2, replace=False)).astype(np.int32)
    s1 = s2 if selfd else np.sort(rng.choice(N, n1, replace=False)).astype(np.int32)
    d1, d2 = torch.as_tensor(s1, device=dev), torch.as_tensor(s2, device=dev)
    P = int(lib.mkamd_dist_count_pairs(n1, n2, int(selfd)))
    out = torch.empty((F, P), device=dev)
    for pbc in (False, True):
        for avoid in tuple(int(a) for a in os.environ.get('PROBE_AVOID', '0,1,3').split(',')):                  # free choice; without the block-per-frame kernel (round 4's choice); the tile kernels only
            ctx.set_dist_kernels(avoid)
            ms = t(lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2, chains.data_ptr(), selfd, pbc, False, out.data_ptr()))
            alg = out.numel() * 4 + (n1 + n2) * 3 * F * 4
            print(f"{n1:5d} x {n2:5d} {'selfdist' if selfd else '        '} pbc={pbc!s:5} avoid={avoid}: {ms * 1e3:8.1f} us  {alg / ms / 1e6:6.0f} GB/s algorithmic "
                  f"({out.numel() * 4 / 1e6:.0f} MB)  {ctx.last_dist_kernel()}")
ctx.set_dist_kernels(0)
