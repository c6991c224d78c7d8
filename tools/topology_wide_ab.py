#!/usr/bin/env python3
"""tools/topology_wide_ab.py -- cfg4-shaped frames of ONE molecule (256 x 30 000 atoms, periodic, 48^3) with and without atoms
whose sigma is wide enough for the exact cut-off fix-up (sigma > 1.81 A: Zn 2.01, Ca 2.31, Na 2.27 ...), through a topology handle and
through the plain call: ms per step in order (k_tail on the critical path) and pipelined, same box and session.  ADVICE r5: a
topology call used to run its fix-up as ONE job per item (a wave walking the whole frame); now one job per (item, wide atom).
    python tools/topology_wide_ab.py            (under rocprofv3 --kernel-trace --stats for k_tail's own duration)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch
from tests.synth import synth_config, grid_origin

dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
F = int(os.environ.get("AB_FRAMES", "256"))
p = synth_config(4, F)
n = int(p["atom_offsets"][1])
o, nv = grid_origin(p["centers"][0], p["boxsize"], 1.0)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
d_xyz, d_offs = t(p["coords"], np.float32), t(p["atom_offsets"], np.int64)
d_org, d_box = t(np.tile(o, (F, 1)), np.float64), t(p["box"], np.float32)
out = torch.empty((F, int(np.prod(nv)), 8), dtype=torch.float32, device=dev)
rng = np.random.default_rng(2)
CONFIGS = (("no wide atom", 0), ("8 ions (Zn / Ca)", 8), ("300 ions (Na+)", 300), ("every 10th atom wide", n // 10))
if os.environ.get("AB_ONLY"):                      # one configuration alone (profiling passes): AB_ONLY=8
    CONFIGS = tuple(c for c in CONFIGS if str(c[1]) == os.environ["AB_ONLY"])
for label, nwide in CONFIGS:
    sig = p["sigmas"][:n].astype(np.float32).copy()
    if nwide:
        idx = rng.choice(n, nwide, replace=False)
        sig[idx, 6] = np.where(np.arange(nwide) % 2 == 0, 2.01, 2.31)        # the metal channel
        sig[idx, 7] = sig[idx, 6]
    d_sig1 = t(sig, np.float32)
    d_rep = d_sig1.repeat(F, 1).contiguous()
    topo = _lib.Topology(ctx, d_sig1, 1.0)
    row = []
    ref = None
    for mode in ("topology", "topology, hits redone inside k_tail", "plain"):
        ctx.set_exact_redo(-1 if "inside" in mode else 0)           # (round 6, late: the hits of a topology call go to k_exact_redo by default)
        for pipelined in (False, True):
            def step():
                if pipelined:
                    ctx.promise_inputs(None)
                if mode.startswith("topology"):
                    batch.voxelize_lattice_torch(d_xyz, d_offs, None, d_org, nv, 1.0, box=d_box, out=out, ctx=ctx, topology=topo)
                else:
                    batch.voxelize_lattice_torch(d_xyz, d_offs, d_rep, d_org, nv, 1.0, box=d_box, out=out, ctx=ctx)
            t_end = time.perf_counter() + 0.5
            while time.perf_counter() < t_end:
                step(); torch.cuda.synchronize(dev)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(30):
                step()
            torch.cuda.synchronize(dev)
            row.append(f"{mode}{' pipelined' if pipelined else ' in order'} {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms")
        chk = float(out.double().sum())
        ref = chk if ref is None else ref
        assert chk == ref, "topology and plain calls differ"
    ctx.set_exact_redo(0)
    topo.close()
    print(f"{label:>22} ({nwide} of {n} atoms): " + " | ".join(row), flush=True)
