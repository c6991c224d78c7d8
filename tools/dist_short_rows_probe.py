import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
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
for F in (1, 8):
    coords = torch.rand((N, 3, F), device=dev) * 66.9
    box = torch.full((3, F), 66.9, device=dev)
    chains = torch.as_tensor((np.arange(N) // 1000).astype(np.int32), device=dev)
    for n1, n2 in ((2000, 30), (5000, 30), (20000, 40), (30, 5000), (40, 20000), (4000, 50)):
        s1 = np.sort(rng.choice(N, n1, replace=False)).astype(np.int32); s2 = np.sort(rng.choice(N, n2, replace=False)).astype(np.int32)
        d1, d2 = torch.as_tensor(s1, device=dev), torch.as_tensor(s2, device=dev)
        out = torch.empty((F, n1 * n2), device=dev)
        row = []
        for pbc in (False, True):
            ms = t(lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2, chains.data_ptr(), False, pbc, False, out.data_ptr()))
            row.append(f"{'pbc' if pbc else 'open'} {ms*1e3:7.1f} us [{ctx.last_dist_kernel()[:60]}]")
        print(f"F={F} {n1} x {n2}: " + " | ".join(row), flush=True)
