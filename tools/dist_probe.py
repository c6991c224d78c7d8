"""tools/dist_probe.py -- what bounds dist_trajectory on the bench workload: the call timed with and without the image shift /
the square root, against plain fills / copies of the same result size (HBM write floor on this box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculekit_amd import _lib
dev = torch.device("cuda", 0)
N, F, n1, n2 = 30000, 2048, 200, 500
rng = np.random.default_rng(4)
coords = torch.rand((N, 3, F), device=dev) * 66.9
box = torch.full((3, F), 66.9, device=dev)
chains = torch.as_tensor((np.arange(N) // 1000).astype(np.int32), device=dev)
d1 = torch.as_tensor(np.sort(rng.choice(N, n1, replace=False)).astype(np.int32), device=dev)
d2 = torch.as_tensor(np.sort(rng.choice(N, n2, replace=False)).astype(np.int32), device=dev)
out = torch.empty((F, n1 * n2), device=dev)
ctx = _lib.default_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for pbc in (False, True):
    for sq in (False, True):
        for selfd in (False,):
            ms = t(lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2, chains.data_ptr(), selfd, pbc, sq, out.data_ptr()))
            print(f"pbc={pbc} squared={sq}: {ms:.4f} ms  {out.numel()*4/ms/1e6:.0f} GB/s of result")
# a square selection with selfdist (MetricSelfDistance; the pair-table kernel): 450 x 450 -> 101 025 pairs
d3 = torch.as_tensor(np.sort(rng.choice(N, 450, replace=False)).astype(np.int32), device=dev)
out2 = torch.empty((F, 450 * 449 // 2), device=dev)
for pbc in (False, True):
    ms = t(lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d3.data_ptr(), 450, d3.data_ptr(), 450, chains.data_ptr(), True, pbc, False, out2.data_ptr()))
    print(f"selfdist pair-table kernel pbc={pbc}: {ms:.4f} ms  {out2.numel()*4/ms/1e6:.0f} GB/s of result")
print(f"torch fill_ of the result: {t(lambda: out.fill_(1.0)):.4f} ms")
src = torch.empty_like(out)
print(f"torch copy_ of the result: {t(lambda: out.copy_(src)):.4f} ms")
print(f"torch mul (read + write): {t(lambda: torch.mul(src, 2.0, out=out)):.4f} ms")
