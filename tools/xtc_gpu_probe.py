"""tools/xtc_gpu_probe.py -- rate of the device XTC decoder (csrc/xtc_gpu.h) on a synthetic cfg4-shaped file (30 000 atoms) and on a
reference-held trajectory with water runs (3PTB head, repeated), for several chunk sizes: kernel time alone (HIP events)."""
import sys, os, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from moleculekit_amd import _lib, xtc
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
d = tempfile.mkdtemp()
p, _, _ = bench.make_workload("cfg4", 64, seed=4001)
N = 30000
nm = np.ascontiguousarray((p["coords"].reshape(64, N, 3) * np.float32(0.1)).transpose(1, 2, 0))
bv = np.zeros((3, 3, 64), np.float32); bv[0, 0] = bv[1, 1] = bv[2, 2] = 6.69
one = os.path.join(d, "one.xtc"); xtc.write_xtc(one, nm, bv, np.zeros(64, np.float32), np.arange(64))
blob = open(one, "rb").read(); syn = os.path.join(d, "syn.xtc")
with open(syn, "wb") as fh:
    for _ in range(64): fh.write(blob)                      # 4096 frames
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "xtc", "3ptb_traj_head.xtc"), "rb").read()
real = os.path.join(d, "real.xtc")
with open(real, "wb") as fh:
    for _ in range(700): fh.write(src)                      # 4200 frames of 4507 atoms
src4 = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "xtc", "4rws_head.xtc"), "rb").read()
real4 = os.path.join(d, "real4.xtc")
with open(real4, "wb") as fh:
    for _ in range(2100 // max(1, xtc.get_xtc_nframes(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "xtc", "4rws_head.xtc")))): fh.write(src4)
lib = _lib.load()
only = os.environ.get("XTC_PROBE_ONLY", "")                  # "syn" / "real": one file (the PMC passes)
for name, fn in (("synthetic 30000 atoms", syn), ("3ptb head (water runs), 4507 atoms", real), ("4rws head (reference writer, solvated: 75338 atoms)", real4)):
    if only and ((only == "syn") != (fn is syn) or fn is real4):
        continue
    na, nf = xtc.get_xtc_natoms(fn), xtc.get_xtc_nframes(fn)
    for n in ((256, 1024, 2048) if fn is real4 else (256, 1024, 2048, 4096)):
        n = min(n, nf)
        sel = np.arange(n, dtype=np.int64)
        t0 = time.perf_counter(); desc, lo, hi, box, tm, st = xtc.chunk_desc(fn, sel, na); t_desc = time.perf_counter() - t0
        raw = torch.zeros(hi - lo + xtc.XTC_PAD, dtype=torch.uint8).pin_memory()
        t0 = time.perf_counter(); _lib._check(lib.mkamd_xtc_copy_bytes(xtc._path(fn), lo, hi, raw.data_ptr(), 0)); t_copy = time.perf_counter() - t0
        d_raw = raw.to(dev); d_desc = torch.as_tensor(desc, device=dev); d_st = torch.empty(n, dtype=torch.int32, device=dev)
        xyz = torch.empty((n, na, 3), dtype=torch.float32, device=dev)
        work = torch.empty(int(lib.mkamd_xtc_decode_work_bytes(n, na)), dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream(dev)
        run = lambda: _lib._check(lib.mkamd_xtc_decode_dev(ctx._h, s.cuda_stream or None, d_raw.data_ptr(), d_desc.data_ptr(), n, na, 10.0, xyz.data_ptr(), d_st.data_ptr(), work.data_ptr(), work.numel()))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        assert int(d_st.abs().sum()) == 0
        print(f"{name}: {n:5d} frames: kernel {ms:8.3f} ms = {n / ms:8.1f} k frames/s ({n * na / ms / 1e3:7.1f} M atoms/s); host: headers {t_desc * 1e3:6.2f} ms, byte copy {t_copy * 1e3:6.2f} ms ({(hi - lo) / 1e6:.0f} MB)")
