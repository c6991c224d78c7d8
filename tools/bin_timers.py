"""Where a wave of k_bin_direct spends its time: wall-clock cycle sums per section (wave 0 of every 64th block), from a profiling
build of the library (hipcc ... -DMK_PHASE_TIMERS -DMK_BIN_TIMERS -o .variants/libmkamd_bintimers.so):

    MKAMD_LIB=.variants/libmkamd_bintimers.so python tools/bin_timers.py [cfg2|cfg1|cfg4]

A load is charged to the section that first WAITS for it (reading the cycle counter does not wait), in-order calls only
(the pipelined ones keep the chain).
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from moleculekit_amd import _lib, batch

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_BATCH[wl]
p, origins, nv = bench.make_workload(wl, B, seed=7)
dev = torch.device("cuda", 0)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
args = (t(p["coords"], np.float32), t(p["atom_offsets"], np.int64), t(p["sigmas"], np.float32), t(origins, np.float64), nv, p["voxelsize"])
ctx = _lib.default_context(0)
ctx.set_direct_binning(1)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3):
    batch.voxelize_lattice_torch(*args, ctx=ctx)
ctx.synchronize()
lib.mkamd_debug_phase_cycles(buf)
n = 5
for _ in range(n):
    batch.voxelize_lattice_torch(*args, ctx=ctx)
ctx.synchronize()
lib.mkamd_debug_phase_cycles(buf)
v = np.array(list(buf), dtype=np.float64)
names = ["set-up: LDS set, barrier, items of the block, class table", "sigma row -> w (one division) + temp word store", "block's class set (LDS)",
         "class id election", "coords -> cell + offset (double)", "rank in cell (atomic round trip)", "record stores (+ spill)", "block set store (barrier)"]
blocks = ((int(p["atom_offsets"][-1]) + 255) // 256 + 63) // 64      # every 64th block is timed
print(f"{wl}: k_bin_direct, share of a wave's wall-clock cycles per section (sum over {n} launches, {blocks} timed blocks each)")
for nm, x in zip(names, v):
    print(f"  {nm:60s} {100 * x / v.sum():5.1f} %   ({x / n / blocks:8.0f} cycles per wave)")
print(f"  total {v.sum() / n / blocks:8.0f} cycles per wave")
