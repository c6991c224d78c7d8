"""tools/xtc_leg_profile.py -- where the host spends its time in iterVoxelizeXTC(decode="gpu") on the cfg4-shaped file"""
import sys, os, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from moleculekit_amd import _lib, batch, xtc
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
base = 64
p, _, _ = bench.make_workload("cfg4", base, seed=4001)
N = 30000; L = float(p["box"][0, 0])
nm = np.ascontiguousarray((p["coords"].reshape(base, N, 3) * np.float32(0.1)).transpose(1, 2, 0))
bv = np.zeros((3, 3, base), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = L * 0.1
sig = np.ascontiguousarray(p["sigmas"][:N], dtype=np.float32)
d = tempfile.mkdtemp()
one = os.path.join(d, "one.xtc"); xtc.write_xtc(one, nm, bv, np.zeros(base, np.float32), np.arange(base))
blob = open(one, "rb").read(); fn = os.path.join(d, "cfg4.xtc")
with open(fn, "wb") as fh:
    for _ in range(16384 // base): fh.write(blob)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def run(nf):
    n = 0; t0 = time.perf_counter(); marks = []
    for idx, feats in batch.iterVoxelizeXTC(fn, sig, p["centers"][0], p["boxsize"], p["voxelsize"], pbc=True, chunk=chunk, ctx=ctx, frames=np.arange(nf), decode="gpu"):
        n += len(idx); del feats; marks.append(time.perf_counter() - t0)
    torch.cuda.synchronize(dev); marks.append(time.perf_counter() - t0)
    return n, marks
lib = _lib.load(); T = {}
def wrap(name):
    f = getattr(lib, name)
    def g(*a):
        t0 = time.perf_counter(); r = f(*a); T.setdefault(name, []).append(round((time.perf_counter() - t0) * 1e3, 2)); return r
    setattr(lib, name, g)
for nme in ("mkamd_xtc_copy_bytes", "mkamd_xtc_decode_dev", "mkamd_xtc_chunk_desc", "mkamd_voxelize_lattice_dev", "mkamd_ctx_synchronize"):
    wrap(nme)
_pm = torch.Tensor.pin_memory
def pm(self, *a, **k):
    t0 = time.perf_counter(); r = _pm(self, *a, **k); T.setdefault("pin_memory", []).append((self.numel() * self.element_size(), round((time.perf_counter() - t0) * 1e3, 2))); return r
torch.Tensor.pin_memory = pm
run(2 * chunk)
print("warm:", T); T.clear()
n, marks = run(16384)
print("yield times (ms):", [round(m * 1e3, 1) for m in marks]); print("timed:", T); T.clear()
pr = cProfile.Profile(); pr.enable(); run(16384); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
