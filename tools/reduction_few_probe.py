#!/usr/bin/env python3
"""tools/reduction_few_probe.py -- dist_trajectory_reduction on device pointers at FEW frames: k_dist_reduction_few (lanes along the
second groups; block -2) against the kernels whose lanes are frames (closest: block 8; generic: -1), per frame count -- where is the
crossover (DRF_MAX_FRAMES)?    python tools/reduction_few_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib
from tools.benchlib.workloads import reduction_workload

dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
t = lambda a: torch.as_tensor(a, device=dev)


def timed(call, reps=20):
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        call(); torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps * 1e3


for G, A in ((200, 15), (600, 8), (60, 15)):
    for F in (1, 2, 4, 8, 12, 16, 32, 64):
        coords, box, atoms, offs, chains, masses = reduction_workload(G, A, F)
        N = coords.shape[0]
        d_c, d_b, d_a, d_o, d_m, d_ch = t(coords), t(box), t(atoms), t(offs), t(masses), t(chains.astype(np.int32))
        P = G * (G - 1) // 2
        out = torch.empty((F, P), device=dev, dtype=torch.float32)
        row = []
        for pbc in (True, False):
            ref = None
            for blk in (-2, 8, -1):
                ctx.set_reduction_block(blk)
                call = lambda: ctx.dist_reduction_dev(d_c, N, F, d_b, d_a, d_o, G, N, d_a, d_o, G, d_ch, d_ch, True, False, pbc, d_m, 0, 0, out)
                us = timed(call)
                got = out.clone()
                same = "" if ref is None or torch.equal(got, ref) else " DIFFERS"
                ref = got if ref is None else ref
                row.append(f"{'pbc' if pbc else 'open'} {blk}: {us:.1f}{same}")
        if F in (1, 8):
            for blk in (-2, -1):                                     # centre of mass on both sides
                ctx.set_reduction_block(blk)
                call = lambda: ctx.dist_reduction_dev(d_c, N, F, d_b, d_a, d_o, G, N, d_a, d_o, G, d_ch, d_ch, True, False, True, d_m, 1, 1, out)
                row.append(f"com {blk}: {timed(call):.1f}")
        ctx.set_reduction_block(0)
        print(f"G={G} A={A} F={F} us per call: " + " | ".join(row), flush=True)
