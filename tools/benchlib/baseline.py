"""The cpu_baseline leg: the oracle (a port of the reference's serial kernel) timed on the GPU box's host cores -- the ONLY place outside
tests/ and smoke() that touches oracle/, and only as the thing reported beside the GPU number."""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np

from .workloads import ROOT, make_config, make_workload

def cpu_baseline(name):
    """Oracle (port of occupancy_utils.pyx:34-61, serial like the reference) on a bounded sample."""
    from oracle import oracle
    from tests.synth import grid_origin
    cfg = int(name[3:])
    p = make_config(name, 1)
    o, nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])
    s, e = p["atom_offsets"][0], p["atom_offsets"][1]
    if cfg in (2, 4):     # ONE whole grid, all its atoms: ~10 s on one core (the step of the GPU bench is 256 of them)
        sample = f"1 item of {name}: all {e - s} atoms, the full {nv[0]}x{nv[1]}x{nv[2]} grid"
        reps = 1
    else:                 # small molecules: whole grids, repeated
        reps = 200 if cfg != 1 else 8
        sample = f"{reps} items of {name} ({e - s} atoms each, full {nv[0]}x{nv[1]}x{nv[2]} grid)"
    centers = oracle.grid_centers(o, nv, p["voxelsize"])
    box = None if p["box"] is None else p["box"][0]
    oracle.calculate_occupancy(centers[:64], p["coords"][s:e], p["sigmas"][s:e], box=box)   # warm
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = oracle.calculate_occupancy(centers, p["coords"][s:e], p["sigmas"][s:e], box=box)
    dt = time.perf_counter() - t0
    listed, granted = os.cpu_count() or 1, host_cores_granted()
    out = {"value": round(reps * centers.shape[0] * 8 / dt / 1e6, 4), "unit": "Mvoxel-channels/s",
           "cores": 1, "kind": "port", "sample": sample, "seconds": round(dt, 2),
           "cpu_model": host_cpu_model(), "cores_listed": listed, "cores_granted": granted,
           "host_cores_available": listed}
    # the reference is serial (no nogil, OpenMP commented out: setup.py:48); for scale, the same sample split over EVERY core
    # the box grants this process (its affinity mask; the C oracle releases the GIL under ctypes): an embarrassingly
    # parallel bound on what this host's CPU could do.  (Fewer threads are timed too: a box may grant logical cores that
    # share physical ones, and the best figure is the one reported -- with the thread count that gave it.)
    if box is None:
        runs = {}
        for nt in sorted({granted, max(1, granted // 2), min(granted, 16)}):
            t0 = time.perf_counter()
            par = oracle.calculate_occupancy_threads(centers, p["coords"][s:e], p["sigmas"][s:e], nt)
            runs[nt] = time.perf_counter() - t0
            assert np.isfinite(par).all()
        nt_best = min(runs, key=runs.get)
        v = centers.shape[0] * 8 / runs[nt_best] / 1e6
        out["all_cores"] = {"value": round(v, 2), "threads": nt_best, "cores_listed": listed, "cores_granted": granted,
                            "value_at_all_granted_cores": round(centers.shape[0] * 8 / runs[granted] / 1e6, 2),
                            "seconds_by_threads": {str(k): round(t, 3) for k, t in sorted(runs.items())},
                            "speedup_over_serial": round(v / out["value"], 1)}
        # beside the baseline, NOT part of it: the library's own host entry point (mkamd_calculate_occupancy_cpu, SURVEY 8b(2): the
        # same contract through a cell list over the atoms) on the same sample -- product code, checked here against the port's result
        try:
            from moleculekit_amd.occupancy_utils import calculate_occupancy_cpu
            c32 = np.ascontiguousarray(p["coords"][s:e], np.float32)
            s64 = np.ascontiguousarray(p["sigmas"][s:e], np.float64)
            c64 = np.ascontiguousarray(centers, np.float64)
            calculate_occupancy_cpu(c64[:64], c32, s64, np.zeros((64, 8), np.float64), n_threads=1)      # (loads the library)
            ent = {}
            for nt in sorted({1, granted}):
                res = np.zeros((c64.shape[0], 8), np.float64)
                t0 = time.perf_counter()
                for _ in range(reps):
                    calculate_occupancy_cpu(c64, c32, s64, res, n_threads=nt)
                ent[nt] = time.perf_counter() - t0
                same = bool(np.array_equal(res, np.asarray(ref).reshape(res.shape)))
            out["library_host_entry"] = {"what": "mkamd_calculate_occupancy_cpu (product code: cell list over the atoms; not the baseline)",
                                         "value_1_thread": round(reps * c64.shape[0] * 8 / ent[1] / 1e6, 2),
                                         "value_all_granted": round(reps * c64.shape[0] * 8 / ent[granted] / 1e6, 2), "threads": granted,
                                         "unit": "Mvoxel-channels/s", "equal_to_the_port_bit_for_bit": same}
        except Exception as ex:                     # noqa: BLE001  (a reported extra: never the reason the line is missing)
            out["library_host_entry"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    return out


def host_cores_granted():
    """Cores this process may run on (its affinity mask / cgroup), as opposed to the cores /proc lists."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:                                                      # a cgroup v2 CPU quota, when there is one, bounds it further
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()
