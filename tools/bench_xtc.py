"""Decode rate of libmkamd.so's XTC reader (host threads), and the end-to-end XTC -> voxel feed rate.
    python tools/bench_xtc.py            (writes nothing; prints one JSON line)
The trajectory is synthesised by repeating the frames of a reference-held fixture (tests/golden/xtc/3ptb_traj_head.xtc:
4507 atoms) -- XTC frames are self-contained records, so concatenating them gives a valid longer file."""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moleculekit_amd import batch, xtc

src = open(os.path.join(ROOT, "tests", "golden", "xtc", "3ptb_traj_head.xtc"), "rb").read()
reps = 400
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "long.xtc")
    with open(fn, "wb") as f:
        for _ in range(reps):
            f.write(src)
    N, F = xtc.get_xtc_natoms(fn), xtc.get_xtc_nframes(fn)
    out = {"atoms": N, "frames": F, "file_MB": round(len(src) * reps / 1e6, 1)}
    for nt in (1, 16, 0):                # 0 = the library's default (up to 64 host threads)
        xtc.read_xtc(fn, nthreads=nt)
        t0 = time.perf_counter(); xtc.read_xtc(fn, nthreads=nt); dt = time.perf_counter() - t0
        out[f"decode_frames_per_s_{nt}thr"] = round(F / dt, 1)
        out[f"decode_Matoms_per_s_{nt}thr"] = round(F * N / dt / 1e6, 1)
    sig = np.where(np.random.default_rng(0).random((N, 8)) < 0.4, 1.7, 0.0).astype(np.float32)
    c0 = xtc.XTCread(fn, frame=0).coords[:, :, 0].mean(0)
    acc = torch.zeros(8, device="cuda", dtype=torch.float64)
    def run(chunk):
        for _, feats in batch.iterVoxelizeXTC(fn, sig, c0, [24, 24, 24], 1.0, pbc=False, chunk=chunk):
            acc.add_(feats.sum(dim=(0, 1), dtype=torch.float64))
        torch.cuda.synchronize()
    # (a chunk is decoded in blocks of 16 frames, one block per host thread at a time: 256 frames keep 16 threads busy,
    #  1024 frames 64)
    for chunk in (256, 1024):
        run(chunk)
        t0 = time.perf_counter(); run(chunk); dt = time.perf_counter() - t0
        out["xtc_to_voxels_frames_per_s" + ("" if chunk == 256 else f"_chunk{chunk}")] = round(F / dt, 1)
print(json.dumps(out))
