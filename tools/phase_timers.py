"""Where a tile wave's time goes: per-phase wall-clock cycle sums of k_voxelize_tiles (class-sorted path), from a
profiling build of the library (hipcc ... -DMK_PHASE_TIMERS -o .variants/libmkamd_phase.so):

    MKAMD_LIB=.variants/libmkamd_phase.so python tools/phase_timers.py [cfg2|cfg3|cfg1|...]
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
kw = dict(box=None if p["box"] is None else t(p["box"], np.float32),
          max_images=1 if p["box"] is None else batch.max_images_per_atom(p["box"], nv, p["voxelsize"]))
ctx = _lib.default_context(0)
ctx.set_tile_items(0)            # the timers live in the wave-per-tile kernel (voxelize_tile), not in k_voxelize_items
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3):
    batch.voxelize_lattice_torch(*args, ctx=ctx, **kw)
ctx.synchronize()
lib.mkamd_debug_phase_cycles(buf)
n = 5
for _ in range(n):
    batch.voxelize_lattice_torch(*args, ctx=ctx, **kw)
ctx.synchronize()
lib.mkamd_debug_phase_cycles(buf)
v = np.array(list(buf)[:6], dtype=np.float64)
names = ["prologue", "traversal 1 (histogram)", "counts -> starts", "traversal 2 (placement)", "pair loops + flushes", "epilogue"]
print(f"{wl}: share of a tile wave's wall-clock cycles per phase, wave-per-tile kernel (sum over {n} launches)")
for nm, x in zip(names, v):
    print(f"  {nm:28s} {100 * x / v.sum():5.1f} %   ({x / n / 1e3:9.1f} k cycles per launch, summed over the timed waves)")
w = np.array(list(buf)[6:8], dtype=np.float64)
print(f"  inside the traversals: issue + wait for the chunk loads {100 * w[0] / v.sum():5.1f} %, cull / histogram / placement {100 * w[1] / v.sum():5.1f} %")
