"""tools/xtc_overlap_probe.py -- what a chunk's upload and device decode cost ALONE and BESIDE the voxelizer (cfg4 shape):
the H2D rate out of pinned memory, the decode kernels' time on an idle chip and while cfg4 steps run on another stream."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from moleculekit_amd import _lib, batch, xtc
ctx = _lib.default_context(0); dev = torch.device("cuda", 0); lib = _lib.load()
base, N, n = 64, 30000, 2048
p, _, _ = bench.make_workload("cfg4", base, seed=4001)
L = float(p["box"][0, 0])
nm = np.ascontiguousarray((p["coords"].reshape(base, N, 3) * np.float32(0.1)).transpose(1, 2, 0))
bv = np.zeros((3, 3, base), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = L * 0.1
d = tempfile.mkdtemp(); one = os.path.join(d, "one.xtc"); xtc.write_xtc(one, nm, bv, np.zeros(base, np.float32), np.arange(base))
blob = open(one, "rb").read(); fn = os.path.join(d, "f.xtc")
with open(fn, "wb") as fh:
    for _ in range(n // base): fh.write(blob)
desc, lo, hi, _, _, _ = xtc.chunk_desc(fn, np.arange(n), N)
h_raw = torch.empty(hi - lo + xtc.XTC_PAD, dtype=torch.uint8, pin_memory=True)
_lib._check(lib.mkamd_xtc_copy_bytes(xtc._path(fn), lo, hi, h_raw.data_ptr(), 0))
d_raw = torch.empty_like(h_raw, device=dev); d_desc = torch.as_tensor(desc, device=dev); d_st = torch.empty(n, dtype=torch.int32, device=dev)
xyz = torch.empty((n, N, 3), dtype=torch.float32, device=dev)
work = torch.empty(int(lib.mkamd_xtc_decode_work_bytes(n, N)), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(device=dev); side2 = torch.cuda.Stream(device=dev)
# the voxelizer's load: cfg4 steps of 256 frames on the main stream
P, _, _ = bench.make_workload("cfg4", 256, seed=4000)
t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a if dt is None else a.astype(dt)), device=dev)
c, off, sg, org, bx = t(P["coords"]), t(P["atom_offsets"]), t(P["sigmas"], np.float32), t(P["centers"] - P["boxsize"] / 2), t(P["box"], np.float32)
nv = np.ceil(P["boxsize"] / P["voxelsize"]).astype(int)
def vox(k):
    for _ in range(k):
        f = batch.voxelize_lattice_torch(c, off, sg, org, nv, float(P["voxelsize"]), box=bx, max_images=1, ctx=ctx); del f
vox(3); torch.cuda.synchronize()
def timed(fn_, stream, busy):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if busy: vox(40)                                       # ~43 ms of tile kernels queued on the main stream
    with torch.cuda.stream(stream):
        e0.record(stream); fn_(stream); e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
h2d = lambda s: d_raw.copy_(h_raw, non_blocking=True)
dec = lambda s: _lib._check(lib.mkamd_xtc_decode_dev(ctx._h, s.cuda_stream, d_raw.data_ptr(), d_desc.data_ptr(), n, N, 10.0, xyz.data_ptr(), d_st.data_ptr(), work.data_ptr(), work.numel()))
for busy in (False, True):
    for name, f in (("H2D of %d MB" % (h_raw.numel() >> 20), h2d), ("decode of %d frames" % n, dec)):
        ms = [timed(f, side, busy) for _ in range(4)][1:]
        extra = "  = %.1f GB/s" % (h_raw.numel() / min(ms) / 1e6) if f is h2d else ""
        print(f"{'beside the voxelizer' if busy else 'alone':>22}: {name}: {', '.join('%.2f' % m for m in ms)} ms{extra}")
# both at once beside the voxelizer (what the stream does: upload of chunk k+1 beside the decode of chunk k)
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize(); vox(40)
with torch.cuda.stream(side): e[0].record(side); h2d(side); e[1].record(side)
with torch.cuda.stream(side2): e[2].record(side2); dec(side2); e[3].record(side2)
torch.cuda.synchronize()
print(f"  both beside the voxelizer: H2D {e[0].elapsed_time(e[1]):.2f} ms, decode {e[2].elapsed_time(e[3]):.2f} ms")
t0 = time.perf_counter(); vox(40); torch.cuda.synchronize(); print(f"  40 cfg4 steps alone: {(time.perf_counter() - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); vox(40)
with torch.cuda.stream(side2): dec(side2); dec(side2); dec(side2)
torch.cuda.synchronize(); print(f"  40 cfg4 steps beside 3 decodes: {(time.perf_counter() - t0) * 1e3:.1f} ms")
