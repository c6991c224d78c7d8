"""Feed rate of batch.iterVoxelizeTrajectory on a cfg4-shaped host trajectory (30 000 atoms, 48^3 grid @ 1 A, periodic):
frames/s seen by a consumer that reduces each chunk on the device (so only the inputs cross PCIe).
    python tools/bench_feed.py [frames] [chunk]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moleculekit_amd import batch
from tests.synth import synth_sigmas

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, L = 30000, 66.9
rng = np.random.default_rng(4)
xyz = rng.uniform(0, L, size=(N, 3, F)).astype(np.float32)
sig = synth_sigmas(rng, N).astype(np.float32)
box = np.full((3, F), L, np.float32)
acc = torch.zeros(8, device="cuda", dtype=torch.float64)

def run():
    for idx, feats in batch.iterVoxelizeTrajectory(xyz, sig, [L / 2] * 3, [48, 48, 48], 1.0, box=box, chunk=chunk):
        acc.add_(feats.sum(dim=(0, 1), dtype=torch.float64))
    torch.cuda.synchronize()

run()
t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
print(json.dumps({"frames": F, "chunk": chunk, "atoms": N, "grid": [48, 48, 48], "frames_per_s": round(F / dt, 1),
                  "input_GBps_over_pcie": round(F * N * 12 / dt / 1e9, 2), "Mvoxel_channels_per_s": round(F * 48 ** 3 * 8 / dt / 1e6, 1)}))
