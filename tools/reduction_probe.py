#!/usr/bin/env python3
"""tools/reduction_probe.py -- dist_trajectory_reduction (closest / closest) on device pointers over a few shapes: does the time per
atom pair depend on the size of the launch (the tail of the last round of blocks), on the group size (padding slots), on the block?
    python tools/reduction_probe.py            (PROBE_BLOCKS=0,4,8,108,-1 to choose the kernels)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib
from tools.benchlib.workloads import reduction_workload

dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
blocks = [int(b) for b in os.environ.get("PROBE_BLOCKS", "0,4,8,108,-1").split(",")]
shapes = [(200, 15, 512), (200, 15, 2048), (100, 15, 512), (200, 9, 512), (400, 7, 512), (50, 15, 200)]
t = lambda a: torch.as_tensor(a, device=dev)
for G, A, F in shapes:
    coords, box, atoms, offs, chains, masses = reduction_workload(G, A, F)
    N = coords.shape[0]
    d_c, d_b, d_a, d_o, d_m, d_ch = t(coords), t(box), t(atoms), t(offs), t(masses), t(chains.astype(np.int32))
    P = G * (G - 1) // 2
    out = torch.empty((F, P), device=dev, dtype=torch.float32)
    npairs = P * A * A * F
    row = []
    for pbc in (True, False):
        for blk in blocks:
            ctx.set_reduction_block(blk)
            call = lambda: ctx.dist_reduction_dev(d_c, N, F, d_b, d_a, d_o, G, N, d_a, d_o, G, d_ch, d_ch, True, False, pbc, d_m, 0, 0, out)
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:
                call(); torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record(); torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / 10
            row.append(f"{'pbc' if pbc else 'open'} blk{blk}: {ms:.4f} ms {npairs / ms / 1e6:.0f} Gp/s")
    ctx.set_reduction_block(0)
    print(f"G={G} A={A} F={F} ({npairs / 1e9:.2f} G atom pairs): " + " | ".join(row), flush=True)
