#!/usr/bin/env python3
"""Throughput of the distance_utils row on MI355X (secondary bench; the headline bench is /bench.py).

dist_trajectory on an HBM-resident trajectory: N atoms x F frames (reference layout [N,3,F]), n1 x n2 atom pairs,
periodic by chain.  Output-bound: algorithmic bytes = 4 B per (frame, pair) written + the selected atoms'
coordinates read once.  Prints one JSON line with Mdist/s, achieved GB/s vs the 8 TB/s HBM peak and the oracle's
single-core rate on a bounded sample.
"""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from moleculekit_amd import _lib
    from oracle import oracle
    N, F, n1, n2 = 30000, 2048, 200, 500
    rng = np.random.default_rng(4)
    dev = torch.device("cuda", 0)
    coords = torch.rand((N, 3, F), device=dev, dtype=torch.float32) * 66.9
    box = torch.full((3, F), 66.9, device=dev, dtype=torch.float32)
    chains = torch.as_tensor((np.arange(N) // 1000).astype(np.int32), device=dev).view(torch.int32)
    s1 = np.sort(rng.choice(N, n1, replace=False)).astype(np.uint32)
    s2 = np.sort(rng.choice(N, n2, replace=False)).astype(np.uint32)
    d1 = torch.as_tensor(s1.astype(np.int32), device=dev); d2 = torch.as_tensor(s2.astype(np.int32), device=dev)
    out = torch.empty((F, n1 * n2), device=dev, dtype=torch.float32)
    ctx = _lib.default_context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2,
                                chains.data_ptr(), False, True, False, out.data_ptr())
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for _ in range(K):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    ndist = F * n1 * n2
    alg = ndist * 4 + (n1 + n2) * 3 * F * 4 + 3 * F * 4
    # parity spot check against the oracle on a slice, and the CPU baseline (1 core) on a bounded sample
    Fs = 64
    csub = coords[:, :, :Fs].contiguous().cpu().numpy(); bsub = box[:, :Fs].contiguous().cpu().numpy()
    t0 = time.perf_counter()
    ref = oracle.dist_trajectory(csub, bsub, s1, s2, chains.cpu().numpy().astype(np.uint32), False, True)
    cpu_s = time.perf_counter() - t0
    got = out[:Fs].cpu().numpy()
    assert np.array_equal(got, ref), "GPU result is not bit-exact with the oracle"
    print(json.dumps({
        "metric": "Mdist/s (dist_trajectory, periodic by chain)", "value": round(ndist / ms / 1e3, 1), "unit": "Mdist/s",
        "ms_per_step": round(ms, 4), "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{N} atoms x {F} frames, {n1} x {n2} pairs", "bit_exact_vs_oracle": True},
        "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(alg / ms / 1e6 / 8000.0, 4), "traffic": None, "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": {"value": round(Fs * n1 * n2 / cpu_s / 1e6, 2), "unit": "Mdist/s", "cores": 1, "kind": "port",
                         "sample": f"first {Fs} frames of the same workload"}}))


if __name__ == "__main__":
    main()
