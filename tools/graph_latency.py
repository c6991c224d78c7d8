#!/usr/bin/env python3
"""tools/graph_latency.py -- VERDICT r5 item 6, the gated experiment: one grid per call (the reference's only call pattern),
device-resident inputs; the three launches of a small call (k_bin_solo -> k_voxelize_tiles_team -> k_tail) captured ONCE as a HIP
graph through the library's own "_dev" entry point on the caller's stream, then replayed -- against the same calls made one by one.
Gate: adopt a graph in the product only if one 64^3 grid takes <= 32 us per call this way (37.3 us in round 5)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from moleculekit_amd import _lib, batch
from tests.synth import grid_origin, synth_config

dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)


def probe(name, p, reps=300):
    o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    n = int(p["atom_offsets"][1])
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
    args = (t(p["coords"][:n], np.float32), t(p["atom_offsets"][:2], np.int64), t(p["sigmas"][:n], np.float32), t(o[None], np.float64), nv, p["voxelsize"])
    out = torch.empty((1, int(np.prod(nv)), 8), dtype=torch.float32, device=dev)
    ref = torch.empty_like(out)
    s = torch.cuda.Stream(dev)
    call = lambda o_: batch.voxelize_lattice_torch(*args, out=o_, ctx=ctx)
    with torch.cuda.stream(s):
        for _ in range(20):
            call(ref)
        s.synchronize()
        plain = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                call(ref)
            s.synchronize()
            plain = min(plain, (time.perf_counter() - t0) / reps * 1e6)
    # capture the same call (its workspaces exist, its class table is warm: nothing allocates or waits inside)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            call(out)
    except Exception as e:                                  # noqa: BLE001
        print(f"{name}: plain {plain:.1f} us per call; capture FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)
        return
    out.zero_()
    g.replay(); torch.cuda.synchronize(dev)
    same = bool(torch.equal(out, ref))
    graph = 1e9
    for _ in range(3):
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize(dev)
        graph = min(graph, (time.perf_counter() - t0) / reps * 1e6)
    # ... and one replay at a time, waited for (a host caller's pattern)
    lat_plain = lat_graph = 1e9
    for _ in range(3):
        with torch.cuda.stream(s):
            t0 = time.perf_counter()
            for _ in range(100):
                call(ref); s.synchronize()
            lat_plain = min(lat_plain, (time.perf_counter() - t0) / 100 * 1e6)
        t0 = time.perf_counter()
        for _ in range(100):
            g.replay(); torch.cuda.synchronize(dev)
        lat_graph = min(lat_graph, (time.perf_counter() - t0) / 100 * 1e6)
    print(f"{name}: back to back  plain {plain:.1f} us   graph replay {graph:.1f} us   |   waited for each  plain {lat_plain:.1f} us   graph {lat_graph:.1f} us   "
          f"(replay bit-identical: {same})", flush=True)


g3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cfg1_3ptb.npz"))
for r in range(2):
    probe("cfg2 grid (50 000 atoms, 64^3)", synth_config(2, 1))
    probe("3PTB pocket (1 639 atoms, 24^3)", dict(coords=g3["coords"], sigmas=g3["sigmas"], atom_offsets=np.array([0, len(g3["coords"])]), centers=g3["center"][None],
                                                  boxsize=g3["boxsize"], voxelsize=float(g3["voxelsize"])))
