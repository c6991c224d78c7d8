#!/usr/bin/env python3
"""bench.py -- Mvoxel-channels/s of the voxel-descriptor hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg1..cfg5|dist|dropin] [--batch B]

A "step" is one pass of the hot path (bin -> scan -> fill -> tile kernel, all on the GPU) over one
batch of synthetic items that is ALREADY resident in HBM; features stay resident in HBM (float32
[B,V,C]).  Default workload = BASELINE.json configs[1] ("cfg2"): independent 50k-atom solvated-protein
systems, 64^3 grid @ 1 A, 8 channels, B systems per GPU per step (SURVEY.md section 8d generators).

N > 1: one process per GPU over RCCL.  Started by the driver under torch.distributed.run, or -- when
invoked as plain `python bench.py --gpus N` -- this file starts the N ranks itself (launch_ranks) and
fails loudly when the node has fewer devices.  The batch of N x B items is sharded with
moleculekit_amd.distributed.ShardedVoxelizer: every rank generates, stages (pinned memory) and keeps only
its own B items (weak scaling), and the timed region holds no collective.  The trivial gather of the
feature tensors is timed separately: `gather_ms` (one padded all-gather after the compute) and
`gather_overlapped_extra_ms` (chunks gathered on a communication stream while the next chunk is computed).
The headline workload is the same at every N (the driver divides the per-N values); north_star's "batched
molecules" case (cfg3) runs through the same sharded path as a secondary leg (`batched_molecules`).
`--dry-run` exercises the launch / rendezvous / shard / gather plumbing on CPU (gloo, stand-in compute);
MKAMD_BENCH_SHARE_DEVICES=1 rehearses the real multi-process path on a box with fewer GPUs than ranks (ranks share
devices; the line carries `config.rehearsal` and is not a scaling measurement).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events recorded around the tile
kernel on the stream it runs on; at N = 1 `other_workloads` carries the same measurement for the other
BASELINE configs; `cpu_baseline` times the oracle (oracle/liboracle.so, a port of the reference's serial
Cython kernel) on a bounded sample of the same workload on 1 host core.  Before the W warm-up steps the workload runs
untimed for `--settle-seconds` (clocks; `config.clock_settle_s`); `sustained` repeats the steps for seconds and
reports the shader clock the device held meanwhile.  stdout carries the JSON line only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP32_VALU_PEAK_TFLOPS = 157.3  # vector FP32 peak (secondary roofline: cfg2/cfg4 are VALU-bound)

# items per GPU per step: big enough that the tile grid is many waves deep (a 32-grid step of cfg2 is only four:
# its tail costs ~15 %) and that the fixed per-step costs (launch gaps, the barrier of the timed region) amortise
# (cfg5's nominal batch is 100 000 items, BASELINE.json configs[4]; the small-molecule workloads keep gaining up to
#  ~32 k grids per step: DESIGN.md section 5)
DEFAULT_BATCH = {"cfg1": 4096, "cfg2": 256, "cfg3": 32768, "cfg4": 256, "cfg5": 65536, "dist": 2048, "dropin": 1}


def real_protein_config(batch, seed):
    """BASELINE configs[0] scaled up: the reference's own 3PTB pocket case (real protein density and
    channel typing, tests/golden/cfg1_3ptb.npz), `batch` randomly rotated copies (rotation about the
    grid centre, what the reference's augmentation loop feeds getVoxelDescriptors)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_3ptb.npz"))
    rng = np.random.default_rng(seed)
    c0 = g["coords"].astype(np.float64) - g["center"][None, :]
    q = rng.normal(size=(batch, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(batch, 3, 3)
    coords = (np.einsum("bij,nj->bni", R, c0) + g["center"][None, None, :]).astype(np.float32)
    n = c0.shape[0]
    return dict(coords=coords.reshape(-1, 3), sigmas=np.tile(g["sigmas"], (batch, 1)),
                atom_offsets=np.arange(batch + 1, dtype=np.int64) * n,
                centers=np.tile(g["center"][None, :], (batch, 1)).astype(np.float64),
                boxsize=np.asarray(g["boxsize"], dtype=np.float64), voxelsize=float(g["voxelsize"]), box=None)


def make_config(name, batch, seed=None):
    from tests.synth import synth_config
    if name == "cfg1":
        return real_protein_config(batch, 1 if seed is None else seed)
    return synth_config(int(name[3:]), batch, seed=seed)


def make_workload(name, batch, seed):
    from tests.synth import grid_origin
    p = make_config(name, batch, seed)
    if os.environ.get("MKAMD_SPATIAL_ORDER", "0") == "1":
        # experiment knob: atoms of every item in spatially coherent order (8 A blocks), like residues / waters in a
        # real topology, instead of the synthetic generator's random order
        co, sg, offs = p["coords"].copy(), p["sigmas"].copy(), p["atom_offsets"]
        for b in range(len(offs) - 1):
            s0, e0 = int(offs[b]), int(offs[b + 1])
            key = np.floor(co[s0:e0] / 8.0).astype(np.int64)
            order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
            co[s0:e0], sg[s0:e0] = co[s0:e0][order], sg[s0:e0][order]
        p["coords"], p["sigmas"] = co, sg
    origins = np.stack([grid_origin(c, p["boxsize"], p["voxelsize"])[0] for c in p["centers"]])
    nv = grid_origin(p["centers"][0], p["boxsize"], p["voxelsize"])[1]
    return p, origins, nv


def pmc_entry(workload, batch, tile_k, kernel=None):
    """Counters of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside the bench):
    profiles/r*_<workload>_pmc_counters.json of THIS build -- the file carries the source hash of the library it was taken
    on (`_library_src`, moleculekit_amd._lib.source_hash()) and counters of another build are refused, with the reason on
    the line.  -> (entry dict | None, file | reason, whole-step HBM bytes | None)"""
    import glob
    from moleculekit_amd import _lib
    if batch != DEFAULT_BATCH[workload] or tile_k not in (0, 8):
        return None, "no pass at this batch / tile depth (the passes are taken at the defaults)", None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc_counters.json")),
                   key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
    if not files:
        return None, "no PMC pass committed for this workload", None
    d = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if d.get("_library_src") != _lib.source_hash():
        return None, f"refused: {rel} was taken on build {d.get('_library_src')}, this is {_lib.source_hash()}", None
    if d.get("_items_per_launch") != batch:
        return None, f"refused: {rel} was taken at another batch", None
    best = d.get(kernel) if kernel and isinstance(d.get(kernel), dict) else None
    if best is None:                               # the instance most launches ran (the LDS tier the host settled on)
        for k, v in d.items():
            if isinstance(v, dict) and ("k_voxelize_tiles<8" in k or "k_voxelize_tiles_lean<8" in k or "k_voxelize_items<8" in k) \
                    and "FETCH_SIZE" in v and "WRITE_SIZE" in v and (best is None or v.get("_launches", 0) > best.get("_launches", 0)):
                best = v
    if best is None or "FETCH_SIZE" not in best or "WRITE_SIZE" not in best:
        return None, f"refused: {rel} holds no counters of {kernel}", None
    # the whole step: every kernel of the pass (pre-pass, tile kernel, tail), launches per step from the launch counts
    step = None
    if d.get("_full_batch_launches_only") and best.get("_launches"):
        step = 0.0
        for k, v in d.items():
            if isinstance(v, dict) and k.startswith("mkamd::") and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                step += (v["WRITE_SIZE"] + 2.0 * v["FETCH_SIZE"]) * 1024 * v.get("_launches", 0) / best["_launches"]
        step = int(step)
    return best, rel, step


def in_range_pairs(p, nv, items):
    """(voxel, atom x channel) pairs within the 5 A cutoff -- what the reference's loop accepts (occupancy_utils.pyx:53) --
    counted exactly on `items` of the workload (host, numpy): every atom against the lattice points of the 11^3 voxels
    around it (periodic items: its images within reach of the grid)."""
    from tests.synth import grid_origin
    vs = float(p["voxelsize"])
    R = 5.0 / vs
    w = int(np.ceil(R)) + 1
    off = np.stack(np.meshgrid(*[np.arange(-w, w + 1)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    total = 0
    for b in items:
        s, e = int(p["atom_offsets"][b]), int(p["atom_offsets"][b + 1])
        o, _ = grid_origin(p["centers"][b], p["boxsize"], vs)
        x = (p["coords"][s:e].astype(np.float64) - o) / vs                      # voxel units, voxel i at i
        nch = (np.asarray(p["sigmas"][s:e]) != 0).sum(1).astype(np.int64)
        if p["box"] is not None:                                                # images that can reach the grid
            L = p["box"][b].astype(np.float64) / vs
            sh = np.stack(np.meshgrid(*[np.arange(-1, 2)] * 3, indexing="ij"), -1).reshape(-1, 3) * L
            x = (x[None] + sh[:, None]).reshape(-1, 3)
            nch = np.tile(nch, 27)
        keep = np.all((x > -R) & (x < np.asarray(nv) - 1 + R), axis=1) & (nch > 0)
        x, nch = x[keep], nch[keep]
        for i in range(0, len(x), 4096):
            xi = x[i:i + 4096]
            base = np.rint(xi)
            pts = base[:, None, :] + off[None]                                  # [n, (2w+1)^3, 3]
            ok = (((pts - xi[:, None]) ** 2).sum(-1) < R * R) & np.all((pts >= 0) & (pts < np.asarray(nv)), axis=-1)
            total += int((ok.sum(1) * nch[i:i + 4096]).sum())
    return total


def algorithmic_flops(p, nv, C=8):
    """SURVEY.md section 8d, secondary roofline: ~22 lane-operations per in-range (voxel, entry) pair (with the min-q
    shortcut) + ~14 per voxel-channel of epilogue (12 operations, 2 transcendentals); the pairs counted on the first items
    of the batch (all of a cfg2 / cfg4 item; 16 small molecules) and scaled to the batch."""
    B = len(p["atom_offsets"]) - 1
    V = int(np.prod(nv))
    items = list(range(min(B, 1 if int(p["atom_offsets"][1]) > 5000 else 16)))
    pairs = in_range_pairs(p, nv, items) / len(items)
    return int(B * (22.0 * pairs + 14.0 * V * C)), pairs / V


def algorithmic_bytes(p, nv, C=8):
    """per grid V*C*4 (one float32 write per voxel-channel) + N*(12 + 4*C)
    (coords + per-channel sigmas read once); summed over the batch."""
    B = len(p["atom_offsets"]) - 1
    V = int(np.prod(nv))
    return B * V * C * 4 + int(p["atom_offsets"][-1]) * (12 + 4 * C)


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


def dist_traffic(F):
    """HBM bytes per step of the distance leg (k_dist_rows and the k_sel_to_frames launch before it; k_dist_rect / k_dist_pairs for builds or shapes that take those) from the committed PMC passes of THIS build (profiles/r*_dist_pmc_counters.json, taken
    at the default frame count; stamped and checked like pmc_entry): WRITE_SIZE + 2 x FETCH_SIZE KiB.  -> (bytes | None, file | reason)"""
    import glob
    from moleculekit_amd import _lib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dist_pmc_counters.json")), key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
    if not files:
        return None, "no PMC pass committed"
    d = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if d.get("_library_src") != _lib.source_hash():
        return None, f"refused: {rel} was taken on build {d.get('_library_src')}, this is {_lib.source_hash()}"
    def find(name):
        return next((v for k, v in d.items() if isinstance(v, dict) and name in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v), None)
    kib = lambda v: v["WRITE_SIZE"] + 2.0 * v["FETCH_SIZE"]
    rows, turn = find("k_dist_rows"), find("k_sel_to_frames")
    if d.get("_items_per_launch") != F:
        return None, f"refused: {rel} was taken at another frame count"
    if rows is not None and turn is not None:               # the row kernel + the launch that turns the selections frame-major
        return int((kib(rows) + kib(turn)) * 1024), rel
    v = find("k_dist_rect") or find("k_dist_pairs")
    if v is None:
        return None, f"refused: {rel} holds no distance-kernel counters"
    return int(kib(v) * 1024), rel


def bench_distances(args, emit=True):
    """Secondary workload (`--workload dist`): `dist_trajectory` on an HBM-resident
    trajectory, 30 000 atoms x F frames (reference layout [N,3,F]), 200 x 500 atom pairs, periodic by chain.
    Output-bound: algorithmic bytes = 4 B per (frame, pair) written + the selected atoms' coordinates read once.
    The cpu_baseline leg times the oracle on the first 64 frames and doubles as a bit-exactness check."""
    import torch
    from moleculekit_amd import _lib
    N, F, n1, n2 = 30000, args.batch or DEFAULT_BATCH["dist"], 200, 500
    rng = np.random.default_rng(4)
    dev = torch.device("cuda", 0)
    coords = torch.rand((N, 3, F), device=dev, dtype=torch.float32) * 66.9
    box = torch.full((3, F), 66.9, device=dev, dtype=torch.float32)
    chains_h = (np.arange(N) // 1000).astype(np.uint32)
    chains = torch.as_tensor(chains_h.astype(np.int32), device=dev)
    s1 = np.sort(rng.choice(N, n1, replace=False)).astype(np.uint32)
    s2 = np.sort(rng.choice(N, n2, replace=False)).astype(np.uint32)
    d1, d2 = torch.as_tensor(s1.astype(np.int32), device=dev), torch.as_tensor(s2.astype(np.int32), device=dev)
    out = torch.empty((F, n1 * n2), device=dev, dtype=torch.float32)
    ctx = _lib.default_context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def busy(call, seconds=0.4):
        """Calls back to back for `seconds` before a timed leg: a leg of a few milliseconds on an idle GPU is timed at the clocks
        it finds (the stand-alone `--workload dist` run read 0.39 / 0.55 of the roofline where the same legs read 0.44 / 0.68 at
        the end of the default run, behind seconds of other work)."""
        if not getattr(args, "settle_seconds", 1.0):   # (--settle-seconds 0: profiling passes, where every launch is a row of the trace)
            return
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            for _ in range(16):
                call()
            torch.cuda.synchronize(dev)

    def timed(pbc):
        def step():
            ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), d1.data_ptr(), n1, d2.data_ptr(), n2,
                                    chains.data_ptr(), False, pbc, False, out.data_ptr())
        busy(step)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # same stream as the kernel
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize(dev)
        elapsed, ms = time.perf_counter() - t0, e0.elapsed_time(e1) / args.steps
        clocks[pbc] = _clock_ghz(ctx, dev, step)     # the shader clock the device holds under THIS leg (beside the roofline fraction it explains)
        return elapsed, ms

    clocks = {}
    only = os.environ.get("MKAMD_DIST_ONLY", "")           # profiling passes: "periodic" / "nonperiodic" / "reduction" = that leg alone (one kernel variant per pass)
    if only == "reduction":                                 # the group-reduction leg alone (periodic, the default block): its PMC / trace passes
        line = {"metric": "G atom pairs/s (dist_trajectory_reduction, closest, periodic; MKAMD_DIST_ONLY=reduction)", "n_gpus": 1, "steps": args.steps,
                "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "reduction": bench_reductions(args, ctx, dev, busy, False, only_periodic=True)}
        line["value"], line["unit"] = line["reduction"]["periodic"]["valu"]["atom_pairs_per_s_G"], "G atom pairs/s"
        line["ms_per_step"] = line["reduction"]["periodic"]["ms_per_call"]
        line["config"] = {"workload": line["reduction"]["shape"]}
        if emit:
            print(json.dumps(line), flush=True)
        return line
    # the common MetricDistance call first (pbc = False: no pair wraps; projections/util.py:30-37), then the headline of this
    # leg, periodic by chain -- whose result stays in `out` for the bit-exactness check below
    ndist = F * n1 * n2
    alg = ndist * 4 + (n1 + n2) * 3 * F * 4 + 3 * F * 4
    nonperiodic = None                                      # (a profiling pass of the periodic leg alone: no numbers are made up for the other)
    if only != "periodic":
        np_elapsed, np_ms = timed(False)
        nonperiodic = {"value": round(ndist * args.steps / np_elapsed / 1e6, 1), "unit": "Mdist/s", "ms_per_step": round(np_elapsed / args.steps * 1e3, 4),
                       "roofline": {"bound": "hbm", "achieved": round(alg / np_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(alg / np_ms / 1e6 / HBM_PEAK_GBS, 4), "kernel": ctx.last_dist_kernel(), "timed_region": "the whole call: " + ctx.last_dist_kernel(), "kernel_avg_ms": round(np_ms, 5),
                                    "shader_clock_ghz": clocks.get(False)}}
    if not args.no_cpu_baseline and only != "periodic":
        from oracle import oracle
        Fs = min(16, F)
        ref = oracle.dist_trajectory(coords[:, :, :Fs].contiguous().cpu().numpy(), box[:, :Fs].contiguous().cpu().numpy(), s1, s2, chains_h, False, False)
        if not np.array_equal(out[:Fs].cpu().numpy(), ref):
            raise SystemExit("dist_trajectory (pbc = False) on the GPU is not bit-exact with the oracle")
    if only == "nonperiodic":                               # profiling pass of the non-periodic leg alone: that leg is the line
        line = {"metric": "Mdist/s (dist_trajectory, pbc = False; MKAMD_DIST_ONLY=nonperiodic)", **nonperiodic, "n_gpus": 1, "steps": args.steps,
                "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"dist: {N} atoms x {F} frames, {n1} x {n2} pairs (SURVEY.md 8f-1)"}, "periodic": None}
        if emit:
            print(json.dumps(line), flush=True)
        return line
    elapsed, k_ms = timed(True)
    line = {"metric": "Mdist/s (dist_trajectory, periodic by chain)", "value": round(ndist * args.steps / elapsed / 1e6, 1),
            "unit": "Mdist/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"dist: {N} atoms x {F} frames, {n1} x {n2} pairs (SURVEY.md 8f-1)"},
            "roofline": {"bound": "hbm", "achieved": round(alg / k_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / k_ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": dist_traffic(F)[0], "traffic_source": dist_traffic(F)[1], "kernel": ctx.last_dist_kernel(), "timed_region": "the whole call: " + ctx.last_dist_kernel() + " (HIP events around the steps)",
                         "kernel_avg_ms": round(k_ms, 5), "algorithmic_bytes_per_launch": alg, "shader_clock_ghz": clocks.get(True)},
            "nonperiodic": nonperiodic}
    if not args.no_cpu_baseline:
        from oracle import oracle
        Fs = min(64, F)
        csub, bsub = coords[:, :, :Fs].contiguous().cpu().numpy(), box[:, :Fs].contiguous().cpu().numpy()
        t0 = time.perf_counter()
        ref = oracle.dist_trajectory(csub, bsub, s1, s2, chains_h, False, True)
        cpu_s = time.perf_counter() - t0
        if not np.array_equal(out[:Fs].cpu().numpy(), ref):
            raise SystemExit("dist_trajectory on the GPU is not bit-exact with the oracle")
        line["cpu_baseline"] = {"value": round(Fs * n1 * n2 / cpu_s / 1e6, 2), "unit": "Mdist/s", "cores": 1, "kind": "port",
                                "sample": f"first {Fs} frames of the same workload (also checked bit-exact)"}
    # ---- the other shapes the projections send (round 5): MetricSelfDistance's triangular list, and the small call MetricDistance
    #      usually makes (protein C-alphas x ligand atoms); each checked bit-exact on its first frames before it is timed ----
    if only == "":
        def shape_leg(sb, sa, selfd, check):                         # sb: first atoms, sa: second atoms
            n1s, n2s = len(sb), len(sa)
            da, db = torch.as_tensor(sb.astype(np.int32), device=dev), torch.as_tensor(sa.astype(np.int32), device=dev)
            Pn = int(lib_count(n1s, n2s, selfd))
            o2 = torch.empty((F, Pn), device=dev, dtype=torch.float32)
            algn = F * Pn * 4 + ((n2s if selfd else n1s + n2s)) * 3 * F * 4 + 3 * F * 4
            res = {}
            for pbc in (False, True):
                call = lambda: ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), da.data_ptr(), n1s, db.data_ptr(), n2s, chains.data_ptr(),
                                                       selfd, pbc, False, o2.data_ptr())
                busy(call, 0.2)
                for _ in range(max(3, args.warmup)):
                    call()
                torch.cuda.synchronize(dev)
                if check:
                    from oracle import oracle
                    Fs = min(8, F)
                    ref = oracle.dist_trajectory(coords[:, :, :Fs].contiguous().cpu().numpy(), box[:, :Fs].contiguous().cpu().numpy(), sb, sa, chains_h, selfd, pbc)
                    if not np.array_equal(o2[:Fs].cpu().numpy(), ref):
                        raise SystemExit(f"dist_trajectory {n1s} x {n2s} selfdist={selfd} pbc={pbc} on the GPU is not bit-exact with the oracle")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    call()
                e1.record()
                torch.cuda.synchronize(dev)
                ms = e0.elapsed_time(e1) / args.steps
                res["periodic" if pbc else "nonperiodic"] = {"us_per_call": round(ms * 1e3, 2), "frac": round(algn / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                             "achieved_GBs": round(algn / ms / 1e6, 1), "kernel": ctx.last_dist_kernel(),
                                                             "algorithmic_bytes_per_launch": algn, "shader_clock_ghz": _clock_ghz(ctx, dev, call)}
            res["shape"] = f"{n1s} x {n2s}{' selfdist' if selfd else ''}: {Pn} pairs x {F} frames ({F * Pn * 4 / 1e6:.0f} MB of result)"
            del o2
            return res
        lib_count = lambda a, b, sd: _lib.load().mkamd_dist_count_pairs(a, b, int(sd))
        check = not args.no_cpu_baseline
        line["selfdist"] = shape_leg(s2[:450].copy(), s2[:450].copy(), True, check)
        line["small_call"] = shape_leg(s2[:300].copy(), s1[:30].copy(), False, check)
        # ---- round 6: the group reductions (MetricDistance's residue contact maps) and the device-side contact lists ----
        line["reduction"] = bench_reductions(args, ctx, dev, busy, check)
        line["contacts"] = bench_contacts(args, ctx, dev, busy, check, coords, box, chains, chains_h, s1, s2, d1, d2, F)
    del coords, out
    torch.cuda.empty_cache()
    if emit:
        print(json.dumps(line), flush=True)
    return line


def _clock_ghz(ctx, dev, call, us=30000):
    """Shader clock the device holds while `call` runs back to back: one wave on a stream of its own counts shader clock ticks against
    the fixed 100 MHz reference for `us` microseconds (mkamd_clock_probe_dev) beside the calls.  GHz, or None."""
    try:
        import torch
        ticks = torch.zeros(2, dtype=torch.int64, device=dev)
        ps = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        ctx.clock_probe_dev(ps.cuda_stream, us, ticks.data_ptr())
        t_end = time.perf_counter() + us * 1.3e-6
        while time.perf_counter() < t_end:
            call()
            torch.cuda.current_stream(dev).synchronize()
        ps.synchronize()
        sh, ref = (int(v) for v in ticks.cpu().tolist())
        return round(sh / ref * 0.1, 3) if ref > 0 else None
    except Exception:                                   # noqa: BLE001  (a reported extra)
        return None


def reduction_workload(G=200, A=15, F=512, L=60.0, seed=5):
    """tools/bench_reduction.py's protein-like trajectory (the shape the 2.9 ms of round 2 were measured on): G residues of A
    atoms (centres uniform in an L^3 box, atoms N(0, 1.5 A) around them, N(0, 0.3 A) per frame), 4 chains of G/4 residues."""
    rng = np.random.default_rng(seed)
    N = G * A
    centres = rng.uniform(0, L, size=(G, 3))
    c0 = (np.repeat(centres, A, axis=0) + rng.normal(0, 1.5, size=(N, 3))).astype(np.float32)
    coords = np.ascontiguousarray((c0[:, :, None] + rng.normal(0, 0.3, size=(N, 3, F))).astype(np.float32))
    box = np.full((3, F), L, dtype=np.float32)
    atoms = np.arange(N, dtype=np.int32)
    offs = (np.arange(G + 1, dtype=np.int64) * A)
    chains = (np.arange(G) // max(1, G // 4)).astype(np.uint32)
    return coords, box, atoms, offs, chains, np.ones(N, np.float32)


def bench_reductions(args, ctx, dev, busy, check, only_periodic=False):
    """dist_trajectory_reduction (distance_utils.pyx:211-281) on device pointers: all 19 900 pairs of 200 residues of 15 atoms,
    512 frames -- 2.29 G atom-pair distances per call.  The path is bound by instruction issue, not by memory (59 MB of
    algorithmic traffic per call): `roofline` is the HBM line the contract asks for, `valu` the one that bounds it
    (atom pairs per second; issue slots the chip had per atom pair at the measured clock; instructions per pair from the
    committed PMC pass of this build when there is one)."""
    import torch
    from moleculekit_amd import _lib
    G, A, F = 200, 15, 512
    coords, box, atoms, offs, chains, masses = reduction_workload(G, A, F)
    N = coords.shape[0]
    t = lambda a: torch.as_tensor(a, device=dev)
    d_c, d_b, d_a, d_o, d_m = t(coords), t(box), t(atoms), t(offs), t(masses)
    d_ch = t(chains.astype(np.int32))
    P = G * (G - 1) // 2
    out = torch.empty((F, P), device=dev, dtype=torch.float32)
    groups = [atoms[offs[g]:offs[g + 1]].tolist() for g in range(G)]
    res = {"shape": f"{G} groups x {A} atoms, {F} frames, all {P} group pairs (selfdist): {P * A * A * F / 1e9:.2f} G atom pairs per call"}
    alg = N * 3 * F * 4 + 3 * F * 4 + F * P * 4

    def leg(pbc, r1, r2, pairs=False, block=0):
        ctx.set_reduction_block(block)
        n_out = G if pairs else P
        o = out if not pairs else torch.empty((F, G), device=dev, dtype=torch.float32)
        call = lambda: ctx.dist_reduction_dev(d_c, N, F, d_b, d_a, d_o, G, N, d_a, d_o, G, d_ch, d_ch, not pairs, pairs, pbc, d_m, r1, r2, o)
        busy(call, 0.3)
        for _ in range(max(3, args.warmup)):
            call()
        torch.cuda.synchronize(dev)
        if check:
            from oracle import oracle
            Fs = 4
            ref = oracle.dist_trajectory_reduction(coords[:, :, :Fs].copy(), box[:, :Fs].copy(), groups, groups, chains, chains, not pairs, pbc, masses,
                                                   r1, r2, pairs=pairs)
            if not np.array_equal(o[:Fs].cpu().numpy(), ref, equal_nan=True):
                raise SystemExit(f"dist_trajectory_reduction pbc={pbc} r=({r1},{r2}) pairs={pairs} block={block} on the GPU is not bit-exact with the oracle")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            call()
        e1.record()
        torch.cuda.synchronize(dev)
        ctx.set_reduction_block(0)
        if want_clock:
            clock[0] = _clock_ghz(ctx, dev, call)
        return e0.elapsed_time(e1) / args.steps

    info = ctx.device_info()
    lanes = info["compute_units"] * 4 * 16
    clock, want_clock = [None], True
    for name, pbc in (("periodic", True), ("nonperiodic", False))[:1 if only_periodic else 2]:
        ms = leg(pbc, 0, 0)
        clk = clock[0]
        npairs = P * A * A * F
        entry = {"ms_per_call": round(ms, 4), "kernel": "mkamd::k_dist_reduction_closest",
                 "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 5),
                              "algorithmic_bytes_per_launch": alg, "note": "instruction-bound: see valu"},
                 "valu": {"atom_pairs_per_s_G": round(npairs / ms / 1e6, 1),
                          "issue_slots_per_atom_pair": None if not clk else round(lanes * clk * 1e9 * ms * 1e-3 / npairs, 2),
                          "shader_clock_ghz": clk}}
        res[name] = entry
    if only_periodic:
        return res
    want_clock = False
    # same-box A-B: the block sizes of the new kernel and the generic kernel it replaces (round 2-5: 2.9 ms on record)
    res["ab_periodic_ms"] = {"block4": round(leg(True, 0, 0, block=4), 4), "block8": round(leg(True, 0, 0, block=8), 4),
                             "block8_four_waves": round(leg(True, 0, 0, block=108), 4), "generic_kernel": round(leg(True, 0, 0, block=-1), 4)}
    res["com_com_periodic_ms"] = round(leg(True, 1, 1), 4)
    res["pairs_closest_periodic_ms"] = round(leg(True, 0, 0, pairs=True), 4)
    pmc = reduction_pmc()
    if pmc:
        res["periodic"]["valu"].update(pmc)
    return res


def reduction_pmc():
    """VALU instructions per atom pair of the periodic leg from the committed PMC pass of THIS build (profiles/r*_reduction_pmc_counters.json)."""
    import glob
    from moleculekit_amd import _lib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reduction_pmc_counters.json")), key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
    if not files:
        return None
    d = json.load(open(files[-1]))
    if d.get("_library_src") != _lib.source_hash():
        return {"pmc": f"refused: {os.path.relpath(files[-1], ROOT)} was taken on build {d.get('_library_src')}, this is {_lib.source_hash()}"}
    v = next((x for k, x in d.items() if isinstance(x, dict) and "k_dist_reduction_closest" in k and "SQ_INSTS_VALU" in x), None)
    if v is None:
        return None
    pairs = 200 * 199 // 2 * 225 * 512
    return {"valu_wave_instructions_per_call": v["SQ_INSTS_VALU"], "lane_instructions_per_atom_pair": round(v["SQ_INSTS_VALU"] * 64 / pairs, 2),
            "pmc_source": os.path.relpath(files[-1], ROOT)}


def bench_contacts(args, ctx, dev, busy, check, coords, box, chains, chains_h, s1, s2, d1, d2, F):
    """contacts_trajectory (distance_utils.pyx:59-93) on device pointers: the dist leg's 200 x 500 pairs over its F frames,
    threshold 8 A, periodic by chain -- counted, scanned and compacted on the device (two passes over the pairs), the list
    stays in HBM.  Algorithmic bytes: the selected atoms' coordinates once + 8 B per contact + the frame offsets."""
    import torch
    n1, n2 = len(s1), len(s2)
    thr = 8.0
    state = {}

    def call():
        state["r"] = ctx.contacts_trajectory_dev(coords, F, box, d1, n1, d2, n2, chains, False, True, thr)
    busy(call, 0.2)
    for _ in range(max(2, args.warmup)):
        call()
    offs, ptr, n = state["r"]
    if check:
        from oracle import oracle
        Fs = min(8, F)
        ref = oracle.dist_trajectory(coords[:, :, :Fs].contiguous().cpu().numpy(), box[:, :Fs].contiguous().cpu().numpy(), s1, s2, chains_h, False, True, squared=True)
        want = [int((ref[f] <= np.float32(thr) * np.float32(thr)).sum()) for f in range(Fs)]
        if want != np.diff(offs[:Fs + 1]).tolist():
            raise SystemExit("contacts_trajectory on the GPU does not count what the oracle counts")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call()
    torch.cuda.synchronize(dev)                      # (every call ends with its own wait for the counts: wall clock = device time + read-backs)
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    alg = (n1 + n2) * 3 * F * 4 + 3 * F * 4 + n * 8 + (F + 1) * 8
    return {"shape": f"{n1} x {n2} pairs x {F} frames, threshold {thr} A, periodic: {n} contacts", "ms_per_call": round(ms, 4),
            "pair_tests_per_s_G": round(n1 * n2 * F / ms / 1e6, 1),
            "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 5),
                         "algorithmic_bytes_per_launch": alg, "note": "instruction-bound: every pair is computed twice (count, fill); wall clock incl. the count read-back"},
            "kernel": "mkamd::k_contacts_count + k_contacts_scan + k_contacts_fill"}


def bench_dropin(args):
    """Secondary workload (`--workload dropin`): the drop-in call itself on BASELINE.json configs[0] (3PTB, 24^3
    grid @ 1 A, 8 channels) -- host numpy arrays in, float64 [V, C] out, every call synchronous, PCIe both ways --
    i.e. what a user who only swaps the import sees; next to the CPU port of the reference loop on the same grid."""
    from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_3ptb.npz"))
    kw = dict(boxsize=[24, 24, 24], center=g["center"], voxelsize=1, usercoords=g["coords"], userchannels=g["sigmas"])
    for _ in range(max(args.warmup, 1)):
        f, c, n = getVoxelDescriptors(None, **kw)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f, c, n = getVoxelDescriptors(None, **kw)
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    V, C = f.shape
    line = {"metric": "Mvoxel-channels/s (drop-in getVoxelDescriptors call, host arrays in/out)",
            "value": round(V * C / ms / 1e3, 2), "unit": "Mvoxel-channels/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "reference fixture (3PTB)",
            "config": {"workload": "dropin: BASELINE.json configs[0], one synchronous call per step, PCIe included"},
            "max_abs_err_vs_reference": float(np.abs(f - g["features"]).max())}
    if not args.no_cpu_baseline:
        from oracle import oracle
        t0 = time.perf_counter()
        oracle.calculate_occupancy(c, g["coords"], g["sigmas"])
        cpu_s = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(V * C / cpu_s / 1e6, 4), "unit": "Mvoxel-channels/s", "cores": 1, "kind": "port",
                                "sample": "the same grid, once", "ms": round(cpu_s * 1e3, 2)}
    print(json.dumps(line), flush=True)


def bench_stream_cfg4(ctx, dev, raw_ms_per_step, frames=2048, chunk=256):
    """Secondary leg `stream_cfg4`: cfg4 frames (30 000 atoms, periodic, 48^3 grid) through the PRODUCT streaming driver
    batch.iterVoxelizeTrajectory from a device-resident trajectory ([N, 3, F], the Molecule.coords layout), `chunk` frames per
    call -- next to the raw pipelined cfg4 step of run_workload (the same 256 frames per call, inputs resident and packed).
    What it adds per chunk: the frame-major transpose on the copy stream, fresh feature memory, the promise / events."""
    import torch
    from moleculekit_amd import batch
    p, _, _ = make_workload("cfg4", chunk, seed=4000)
    N = int(p["atom_offsets"][1])
    src = torch.as_tensor(p["coords"].reshape(chunk, N, 3)).to(dev).permute(1, 2, 0).contiguous().repeat(1, 1, frames // chunk)
    box = np.tile(np.ascontiguousarray(p["box"].T), (1, frames // chunk))
    sig = np.ascontiguousarray(p["sigmas"][:N], dtype=np.float32)

    def run():
        n, marks = 0, []
        for idx, feats in batch.iterVoxelizeTrajectory(src, sig, p["centers"][0], p["boxsize"], p["voxelsize"], box=box, chunk=chunk, ctx=ctx):
            n += len(idx)
            del feats
            ev = torch.cuda.Event(enable_timing=True)      # behind this call's tile kernel on the consumer's stream
            ev.record()
            marks.append(ev)
        torch.cuda.synchronize(dev)
        return n, marks

    run()
    n0 = ctx.pipelined_calls()
    t0 = time.perf_counter()
    n, marks = run()
    dt = time.perf_counter() - t0
    ms_chunk = dt / (n / chunk) * 1e3
    # the cadence of the calls once the pipeline is full (first call's pre-pass, the generator's set-up and its final wait aside)
    steady = marks[0].elapsed_time(marks[-1]) / (len(marks) - 1) if len(marks) > 1 else None
    V = int(np.prod(np.ceil(p["boxsize"] / p["voxelsize"]).astype(int)))
    out = {"frames": n, "frames_per_call": chunk, "frames_per_s": round(n / dt, 1), "ms_per_call": round(ms_chunk, 4),
           "steady_ms_per_call": round(steady, 4) if steady else None,
           "value": round(n * V * 8 / dt / 1e6, 2), "unit": "Mvoxel-channels/s", "pipelined_calls": ctx.pipelined_calls() - n0,
           "source": "device-resident [N,3,F] float32 tensor", "driver": "batch.iterVoxelizeTrajectory (promised inputs, include/mkamd_voxel.h)",
           "note": "ms_per_call = the whole pass (generator set-up, the first call's exposed pre-pass, the final wait) / calls; "
                   "steady_ms_per_call = HIP events behind consecutive calls"}
    if raw_ms_per_step:
        out["raw_pipelined_cfg4_ms_per_step"] = raw_ms_per_step
        out["over_raw_step"] = round(ms_chunk / raw_ms_per_step, 4)
        if steady:
            out["steady_over_raw_step"] = round(steady / raw_ms_per_step, 4)
    del src
    torch.cuda.empty_cache()
    return out


def bench_xtc_cfg4(ctx, dev, raw_ms_per_step, frames=2048, chunk=256, frames_gpu=16384, chunk_gpu=1024):
    """Secondary leg `xtc_cfg4`: the cfg4 FEEDER -- a synthetic 30 000-atom XTC trajectory (64 frames of the cfg4 random walk
    written with moleculekit_amd.xtc.write_xtc, the records repeated: XTC frames are self-contained) voxelized through
    batch.iterVoxelizeXTC, with the coordinates decompressed ON THE DEVICE (decode="auto": csrc/xtc_gpu.h, large chunks) and,
    beside it, by libmkamd.so's host threads (decode="host", the round-3 path).  Reports the end-to-end rates, the host
    decoder's rate alone and how idle the GPU is (the voxelizer's share of the wall time at the raw cfg4 step)."""
    import tempfile
    import torch
    from moleculekit_amd import _lib, batch, xtc
    base = 64
    p, _, _ = make_workload("cfg4", base, seed=4001)
    N = int(p["atom_offsets"][1])
    L = float(p["box"][0, 0])
    nm = np.ascontiguousarray((p["coords"].reshape(base, N, 3) * np.float32(0.1)).transpose(1, 2, 0))      # [N,3,F] in nm
    bv = np.zeros((3, 3, base), np.float32)
    bv[0, 0] = bv[1, 1] = bv[2, 2] = L * 0.1
    sig = np.ascontiguousarray(p["sigmas"][:N], dtype=np.float32)
    per_frame_s = raw_ms_per_step * 1e-3 / DEFAULT_BATCH["cfg4"] if raw_ms_per_step else None
    with tempfile.TemporaryDirectory() as d:
        one = os.path.join(d, "one.xtc")
        xtc.write_xtc(one, nm, bv, np.zeros(base, np.float32), np.arange(base))
        blob = open(one, "rb").read()
        fn = os.path.join(d, "cfg4.xtc")
        with open(fn, "wb") as fh:
            for _ in range(max(frames, frames_gpu) // base):
                fh.write(blob)
        xtc.read_xtc_frames(fn, np.arange(chunk))                               # warm (page cache, threads)
        t0 = time.perf_counter()
        xtc.read_xtc_frames(fn, np.arange(frames))
        t_dec = time.perf_counter() - t0

        def run(decode, nframes, nchunk):
            n, marks = 0, []
            for idx, feats in batch.iterVoxelizeXTC(fn, sig, p["centers"][0], p["boxsize"], p["voxelsize"], pbc=True, chunk=nchunk, ctx=ctx,
                                                    frames=np.arange(nframes), decode=decode):
                n += len(idx)
                ev = torch.cuda.Event(enable_timing=True)                       # when this chunk's features are complete on the device
                ev.record(torch.cuda.current_stream(dev))
                marks.append((ev, n))
                del feats
            torch.cuda.synchronize(dev)
            return n, marks

        def leg(decode, nframes, nchunk):
            run(decode, nframes, nchunk)                                        # warm: buffers, pinned staging, the allocator's blocks
            t0 = time.perf_counter()
            n, marks = run(decode, nframes, nchunk)
            dt = time.perf_counter() - t0
            o = {"frames": n, "frames_per_call": nchunk, "frames_per_s": round(n / dt, 1), "Matoms_per_s": round(n * N / dt / 1e6, 1)}
            if len(marks) >= 4:                                                 # the feed once it is full: chunk 2's features complete
                (ea, na), (eb, nb) = marks[1], marks[-1]                        # -> the last chunk's complete (device events)
                o["steady_frames_per_s"] = round((nb - na) / (ea.elapsed_time(eb) * 1e-3), 1)
            if per_frame_s:
                o["gpu_busy_fraction"] = round(n * per_frame_s / dt, 4)         # the voxelizer's share of the wall time
            return o, dt

        host, _ = leg("host", frames, chunk)
        torch.cuda.empty_cache()
        try:
            gpu, _ = leg("auto", frames_gpu, chunk_gpu)
            # the decode kernels alone, on resident bytes (what one chunk costs beside the voxelizer)
            sel = np.arange(chunk_gpu, dtype=np.int64)
            desc, lo, hi, _, _, _ = xtc.chunk_desc(fn, sel, N)
            raw = torch.from_numpy(np.fromfile(fn, dtype=np.uint8, count=hi - lo, offset=lo))
            d_raw = torch.cat([raw, torch.zeros(xtc.XTC_PAD, dtype=torch.uint8)]).to(dev)
            d_desc = torch.as_tensor(desc, device=dev)
            d_st = torch.empty(chunk_gpu, dtype=torch.int32, device=dev)
            xyz = torch.empty((chunk_gpu, N, 3), dtype=torch.float32, device=dev)
            lib = _lib.load()
            work = torch.empty(int(lib.mkamd_xtc_decode_work_bytes(chunk_gpu, N)), dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream(dev)
            dec = lambda: _lib._check(lib.mkamd_xtc_decode_dev(ctx._h, st.cuda_stream or None, d_raw.data_ptr(), d_desc.data_ptr(), chunk_gpu, N,
                                                               10.0, xyz.data_ptr(), d_st.data_ptr(), work.data_ptr(), work.numel()))
            dec()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); dec(); dec(); e1.record(st)
            torch.cuda.synchronize(dev)
            gpu["decode_kernels_ms_per_call"] = round(e0.elapsed_time(e1) / 2, 3)
            gpu["decode_kernels_frames_per_s"] = round(chunk_gpu / (e0.elapsed_time(e1) / 2) * 1e3, 1)
            del d_raw, d_desc, d_st, xyz, work
        except Exception as e:                     # noqa: BLE001
            gpu = {"error": f"{type(e).__name__}: {e}"[:300]}
    out = {"atoms": N, "file_MB": round(len(blob) * (max(frames, frames_gpu) // base) / 1e6, 1), "bytes_per_atom": round(len(blob) / base / N, 2),
           "frames_per_s": gpu.get("frames_per_s", host["frames_per_s"]), "frames_per_call": gpu.get("frames_per_call", chunk),
           "decode": "device (k_xtc_scan + k_xtc_expand)" if "frames_per_s" in gpu else "host threads",
           "device_decode": gpu,
           "host_decode": dict(host, decode_frames_per_s=round(frames / t_dec, 1), decode_Matoms_per_s=round(frames * N / t_dec / 1e6, 1),
                               host_threads="automatic (<= 64)"),
           "driver": "batch.iterVoxelizeXTC: headers + record bytes (device decode) or decoded coordinates (host decode) -> pinned staging -> "
                     "copy stream -> promised voxelize call"}
    if per_frame_s:
        out["kernels_alone_frames_per_s"] = round(1.0 / per_frame_s, 1)
        fps = out["frames_per_s"]
        out["gpu_busy_fraction"] = round(fps * per_frame_s, 4)
        out["gpu_idle_fraction"] = round(1.0 - fps * per_frame_s, 4)
        out["vs_kernels_alone"] = round(fps * per_frame_s, 4)
        steady = gpu.get("steady_frames_per_s")
        if steady:
            out["steady_vs_kernels_alone"] = round(steady * per_frame_s, 4)
        out["bottleneck"] = ("GPU (voxelizer)" if (steady or fps) * per_frame_s > 0.9 else
                             ("device XTC walk (one lane per frame)" if "frames_per_s" in gpu else "host XTC decode"))
    torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _json_only_stdout():
    """The contract is ONE JSON line on stdout, and libraries write there too (gloo: "[Gloo] Rank n is connected to ..." from every
    rank; RCCL with NCCL_DEBUG set).  From here on file descriptor 1 IS stderr for everything below Python; `print` keeps the real
    stdout through a duplicate.  (Rank processes only: the launcher's children inherit its descriptors.)"""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(keep, "w", buffering=1)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves -- one process per
    GPU under torch.distributed.run on this node (rendezvous on 127.0.0.1) -- and let rank 0 print the line."""
    import subprocess
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and not (os.environ.get("MKAMD_BENCH_SHARE_DEVICES", "0") == "1" and have > 0):
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


class _StandInContext:
    """What run_workload asks of a device context, for --dry-run: there is no CPU voxelizer in the product, so the CPU
    check of the N > 1 path runs the REAL run_workload (sharding, staging, timed loop, fences, max over ranks, both gathers)
    around a stand-in compute and this stand-in context."""

    device = -1

    def synchronize(self):
        pass

    def enable_kernel_timing(self, on):
        pass

    def read_kernel_timing(self):
        return 0.0, 0

    def set_pipelining(self, on):
        pass

    def promise_inputs(self, event=None):
        pass


def _standin_compute(c, offs, s, o, nvox, vs, bx):
    """Stand-in for the HIP path (dry run only): every voxel-channel of an item = 1 + its atom count + 1e-3 x the sum of
    its coordinates -- a function of the ITEM alone, so any chunking / sharding / gather mistake shows up as a wrong row."""
    import torch
    n = len(offs) - 1
    csum = np.concatenate([[0.0], np.cumsum(c.astype(np.float64).sum(axis=1))])
    val = 1.0 + np.diff(offs) + 1e-3 * (csum[offs[1:]] - csum[offs[:-1]])
    return torch.from_numpy(np.broadcast_to(val[:, None, None], (n, int(np.prod(nvox)), s.shape[1])).astype(np.float32).copy())


def dry_run(args):
    """CPU-only check of the N>1 plumbing (tests/test_bench_launch.py): the ranks rendezvous over gloo and go through
    run_workload itself -- the function the timed GPU run uses -- on a small cfg3 batch with a stand-in compute; the
    gathered tensors are then checked row by row against what every rank says its items are worth."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch or 6
    dev = torch.device("cpu")

    def fence():
        _collective(dist.barrier)

    once = threading.Lock()

    def report_degraded():
        if not once.acquire(blocking=False):
            return
        print(json.dumps({"metric": "dry-run (gloo, stand-in compute)", "dry_run": True, "timed_path": "run_workload", "n_gpus": world,
                          "ok": False, "ranks_alive": ranks_done(world), "degraded": _RANKS["broken"]}), flush=True)

    if rank == 0:
        _RANKS["emit"] = report_degraded
        term_reporter()
    t0 = time.perf_counter()
    res = run_workload("cfg3", B, 2, 1, _StandInContext(), dev, rank, world, args, fence, want_gather=not os.environ.get("MKAMD_BENCH_KILL_RANK"),
                       compute=_standin_compute, keep=("full_plain", "full_overlapped", "out"))
    if _RANKS["broken"]:                                      # a rank died: rank 0 says so on its line (the N > 1 GPU run does the same)
        if rank == 0:
            report_degraded()
        os._exit(1)
    mine = res["out"][:, 0, 0].tolist()                       # what this rank's items are worth
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    want = torch.tensor([v for part in everyone for v in part], dtype=torch.float32)
    ok = ("gather_error" not in res and res["k_n"] == 0 and len(mine) == B and min(mine) > 1.0
          and all(tuple(res[k].shape) == (world * B, 24 ** 3, 8) and torch.equal(res[k][:, 0, 0], want)
                  and bool((res[k] == res[k][:, :1, :1]).all()) for k in ("full_plain", "full_overlapped")))
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    # the device every rank WOULD drive (main(): torch.cuda.set_device(LOCAL_RANK), one process per GPU): what the launcher set
    places = [None] * world
    dist.all_gather_object(places, [rank, int(os.environ.get("LOCAL_RANK", "-1"))])
    if rank == 0:
        print(json.dumps({"metric": "dry-run (gloo, stand-in compute)", "dry_run": True, "timed_path": "run_workload", "n_gpus": world,
                          "ranks_joined": world, "ranks_alive": ranks_done(world), "scaling": "weak", "rank_to_local_device": places,
                          "items_per_rank": B, "ok": all(flags), "ms_per_step": round(res["elapsed"] / 2 * 1e3, 3),
                          "gather_ms": res.get("gather_ms"), "gather_error": res.get("gather_error"),
                          "seconds": round(time.perf_counter() - t0, 3)}), flush=True)
    dist.destroy_process_group()
    if not all(flags):
        raise SystemExit("dry run: sharding / gather mismatch")


def _standin_distances(kind, coords, box, sel1, sel2, chains, selfdist, pbc):
    """Stand-in for the distance kernels (dry run only): every pair of a frame = 1 + the frame's coordinate sum -- a function of
    the FRAME alone, so a misplaced or missing row of the sharded / gathered result shows."""
    P = len(sel1) * len(sel2)
    val = 1.0 + coords.astype(np.float64).sum(axis=(0, 1))
    return np.broadcast_to(val[:, None], (coords.shape[2], P)).astype(np.float32).copy()


def bench_distances_sharded(args, dry=False):
    """`--workload dist --gpus N`: dist_trajectory with the FRAMES sharded over the ranks (moleculekit_amd.distributed.ShardedDistances,
    SURVEY.md section 8f-1: "frames shard across GPUs exactly like cfg4").  Weak scaling: F frames per rank, each rank generates and keeps
    only its own; selections replicated; the timed region has no collective; results stay sharded [F_rank, n_pairs].  The gather of the
    rows is a leg of its own after everything timed.  `dry`: gloo + a stand-in compute on CPU tensors (tests/test_bench_launch.py)."""
    import torch
    import torch.distributed as dist
    from moleculekit_amd.distributed import ShardedDistances
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    use_dist = world > 1 or "RANK" in os.environ
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if dry:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev, N, F, n1, n2 = torch.device("cpu"), 50, args.batch or 5, 4, 6
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
        shared = os.environ.get("MKAMD_BENCH_SHARE_DEVICES", "0") == "1" and torch.cuda.device_count() > 0
        if shared:
            local = local % torch.cuda.device_count()
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if use_dist:
            import datetime
            dist.init_process_group("cpu:gloo,cuda:nccl", timeout=datetime.timedelta(seconds=120))
        N, F, n1, n2 = 30000, args.batch or DEFAULT_BATCH["dist"], 200, 500

    def fence():
        if not dry:
            torch.cuda.synchronize(dev)
        if use_dist:
            _collective(lambda: dist.all_reduce(torch.zeros(1)))
        if not dry:
            torch.cuda.synchronize(dev)

    once = threading.Lock()

    def degraded_line():
        if once.acquire(blocking=False):
            print(json.dumps({"metric": "Mdist/s (dist_trajectory, periodic by chain; frames sharded)", "value": None, "unit": "Mdist/s", "n_gpus": world,
                              "dry_run": dry, "ok": False, "ranks_alive": ranks_done(world), "degraded": _RANKS["broken"]}), flush=True)

    if use_dist and rank == 0:
        _RANKS["emit"] = degraded_line
        term_reporter()
    rng = np.random.default_rng(4)                             # the selections: the same on every rank
    chains_h = (np.arange(N) // max(1, N // 30)).astype(np.uint32)
    s1 = np.sort(rng.choice(N, n1, replace=False)).astype(np.uint32)
    s2 = np.sort(rng.choice(N, n2, replace=False)).astype(np.uint32)

    def loader(lo, hi):                                        # this rank's frames (their own seed): nobody builds the whole trajectory
        assert (lo, hi) == (rank * F, (rank + 1) * F)
        r = np.random.default_rng(9000 + rank)
        return (r.random((N, 3, F), dtype=np.float32) * np.float32(66.9)), np.full((3, F), 66.9, np.float32)

    kw = dict(compute=_standin_distances) if dry else dict(device=dev)
    sd = ShardedDistances.from_loader(world * F, loader, **kw)
    out = torch.empty((F, n1 * n2), dtype=torch.float32, device=dev)
    step = lambda: sd.dist_trajectory(s1, s2, chains_h, False, True, out=out)
    steps, warm = (2, 1) if dry else (args.steps, args.warmup)
    if not dry and getattr(args, "settle_seconds", 1.0):
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            for _ in range(16):
                step()
            torch.cuda.synchronize(dev)
    for _ in range(warm):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if not dry:
        torch.cuda.synchronize(dev)
    mark_done(rank)
    fence()
    elapsed = _max_over_ranks(time.perf_counter() - t0, world)
    if _RANKS["broken"]:
        if rank == 0:
            degraded_line()
        os._exit(1)
    line = {"metric": "Mdist/s (dist_trajectory, periodic by chain; frames sharded over the ranks)", "value": round(world * F * n1 * n2 * steps / elapsed / 1e6, 1),
            "unit": "Mdist/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": dry,
            "config": {"workload": f"dist: {N} atoms x {F} frames per rank, {n1} x {n2} pairs (SURVEY.md 8f-1)", "sharding": "contiguous frame ranges, no collective on the compute path"},
            "ranks_alive": ranks_done(world) if use_dist else 1}
    # the gather of the rows (after everything timed: the first RCCL collective of the process builds the communicator)
    if use_dist and not args.no_gather:
        def legs():
            fence()
            g0 = time.perf_counter()
            full = sd.gather(out)
            fence()
            line["gather_ms"] = round((time.perf_counter() - g0) * 1e3, 3)
            root = sd.gather(out, dst=0)
            fence()
            return full, root
        try:
            full, root = guarded(legs, args.gather_timeout, lambda: (degraded_line(), os._exit(1)) if rank == 0 else os._exit(1))
            ok = tuple(full.shape) == (world * F, n1 * n2) and torch.equal(full[rank * F:(rank + 1) * F], out)
            ok = ok and ((root is None) if rank else torch.equal(root, full))
            if dry:                                            # every rank's rows are worth what that rank says they are
                mine = out[:, 0].tolist()
                everyone = [None] * world
                dist.all_gather_object(everyone, mine)
                ok = ok and torch.equal(full[:, 0], torch.tensor([v for part in everyone for v in part], dtype=torch.float32)) and bool((full == full[:, :1]).all())
            flags = [None] * world
            dist.all_gather_object(flags, bool(ok))
            line["gather_ok"] = all(flags)
        except Exception as e:                                 # noqa: BLE001 -- a secondary leg
            line["gather_error"] = f"{type(e).__name__}: {e}"[:300]
    line["ok"] = bool(line.get("gather_ok", True)) and "gather_error" not in line
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        try:
            dist.destroy_process_group()
        except Exception:                                      # noqa: BLE001
            pass
    if dry and not line["ok"]:
        raise SystemExit("dry run: frame sharding / gather mismatch")
    return line


def guarded(fn, seconds, on_timeout):
    """fn() under a watchdog: when it has not returned after `seconds`, on_timeout() is called from another thread (fn
    itself keeps running: a collective that hangs cannot be cancelled, only left behind)."""
    done = threading.Event()

    def watchdog():
        if not done.wait(seconds):
            on_timeout()

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        return fn()
    finally:
        done.set()


# ---- a rank that dies must not cost rank 0 its line ----------------------------------------------------------------------
# Nothing here has run with N > 1 on RCCL where this file was written.  Every collective BETWEEN the ranks (the gloo fences,
# the max over ranks, the gather legs) goes through _collective(): the first failure is remembered, nothing is attempted
# after it, and rank 0 reports what it measured itself with `ranks_alive` (ranks that finished the timed region, read
# from the rendezvous store, which lives in the launcher) and `degraded` on the line.  torchrun answers a dead worker by
# sending the others SIGTERM: rank 0 turns that into its line too (term_reporter: a wake-up fd and a thread, so that it
# works while the main thread sits inside a collective).
_RANKS = {"broken": None, "emit": None}


def _collective(fn, default=None):
    if _RANKS["broken"]:
        return default
    try:
        return fn()
    except Exception as e:                                    # noqa: BLE001 -- reported on the line
        _RANKS["broken"] = f"{type(e).__name__}: {e}"[:200]
        return default


def _store():
    try:
        import torch.distributed as dist
        return dist.distributed_c10d._get_default_store() if dist.is_initialized() else None
    except Exception:                                         # noqa: BLE001
        return None


def mark_done(rank):
    st = _store()
    if st is not None:
        try:
            st.set(f"mkamd_bench_timed_{rank}", "1")
        except Exception:                                     # noqa: BLE001
            pass


def ranks_done(world):
    st = _store()
    if st is None:
        return world
    n = 0
    for r in range(world):
        try:
            n += bool(st.check([f"mkamd_bench_timed_{r}"]))
        except Exception:                                     # noqa: BLE001
            pass
    return n


def term_reporter():
    """SIGTERM -> whatever _RANKS['emit'] holds is called (rank 0's line, as far as it got), then the process ends."""
    import select
    import signal
    r, w = os.pipe()
    os.set_blocking(w, False)
    signal.signal(signal.SIGTERM, lambda *_: None)            # (a Python-level handler must exist for the wake-up fd to fire)
    signal.set_wakeup_fd(w, warn_on_full_buffer=False)

    def watch():
        while True:
            select.select([r], [], [])
            if signal.SIGTERM in os.read(r, 64):
                _RANKS["broken"] = _RANKS["broken"] or "SIGTERM: the launcher is taking the job down (a rank failed)"
                try:
                    if _RANKS["emit"]:
                        _RANKS["emit"]()
                finally:
                    os._exit(1)

    threading.Thread(target=watch, daemon=True).start()


def _max_over_ranks(x, world):
    """MAX of a host scalar over the ranks through a CPU tensor (the gloo side of the process group): nothing in or around
    the timed region touches RCCL -- once an RCCL communicator exists in the process every kernel of the step runs 3-7 %
    slower (measured with one rank, profiles/r3_torchrun_probe.txt), so it is first created by the gather legs, after
    everything that is timed."""
    if world > 1 or "RANK" in os.environ:
        import torch
        import torch.distributed as dist
        tt = torch.tensor([x], dtype=torch.float64)
        return _collective(lambda: (dist.all_reduce(tt, op=dist.ReduceOp.MAX), float(tt.item()))[1], default=x)
    return x


def run_workload(name, B, steps, warmup, ctx, dev, rank, world, args, fence, want_gather=False, want_single=False,
                 compute=None, keep=(), sustain=False, defer_gather=False):
    """Time `steps` passes of the hot path over this rank's resident shard of a batch of world x B items of workload
    `name` (weak scaling).  The shard lives in a moleculekit_amd.distributed.ShardedVoxelizer: loaded by this rank
    alone, staged through pinned memory, resident in HBM before the timed region; no collective inside it.
    `compute` / `keep`: --dry-run only (a stand-in compute on CPU tensors; tensors to hand back for checking)."""
    import torch
    from moleculekit_amd.distributed import ShardedVoxelizer
    cfgno = int(name[3:])
    cache = {}

    def loader(lo, hi):                                       # this rank's B items (their own seed): nobody builds the whole batch
        assert (lo, hi) == (rank * B, (rank + 1) * B)
        p, origins, nv = make_workload(name, B, seed=1000 * cfgno + rank)
        cache.update(p=p, origins=origins, nv=nv)
        sig = np.ascontiguousarray(p["sigmas"], dtype=np.float32)
        return p["coords"], p["atom_offsets"], sig, origins, p["box"]

    from tests.synth import grid_origin
    p0 = make_config(name, 1)
    nv = grid_origin(p0["centers"][0], p0["boxsize"], p0["voxelsize"])[1]
    if compute is None:
        # (pipelined steps are the package's own behaviour now: ShardedVoxelizer promises its resident shard to every call)
        # cfg4 is a trajectory: every item is a frame of ONE molecule, and the package's frame drivers reuse what the pre-pass derives
        # from its sigmas (a topology handle, include/mkamd_voxel.h (3c)); the other workloads are batches of different molecules
        # (cfg1 x 4096 is the augmentation loop: 4 096 rotated copies of ONE pocket -- the same molecule too)
        shared = name in ("cfg1", "cfg4") and not getattr(args, "no_topology", False) and os.environ.get("MKAMD_SPATIAL_ORDER", "0") != "1"
        sv = ShardedVoxelizer.from_loader(world * B, loader, nv, p0["voxelsize"], device=dev, ctx=ctx,
                                          pipelined=not getattr(args, "no_pipeline", False), shared_sigmas=shared)
    else:
        sv = ShardedVoxelizer.from_loader(world * B, loader, nv, p0["voxelsize"], device=dev, compute=compute)
    p = cache["p"]
    V, C = int(np.prod(nv)), 8
    out = torch.empty((B, V, C), dtype=torch.float32, device=dev)
    step = lambda: sv.voxelize(out=out)

    def single_probe():
        """Latency of ONE grid per call (the reference's call pattern; SURVEY.md section 7 H2: a single 64^3 grid cannot fill
        256 CUs for long): device-resident inputs, calls back to back, five runs of 40 calls -> (median, runs) in us."""
        from moleculekit_amd import batch
        d = sv._d
        o1 = torch.empty((1, V, C), dtype=torch.float32, device=dev)
        n1 = int(p["atom_offsets"][1])
        offs1 = d["offs"][:2].contiguous()
        args1 = (d["coords"][:n1], offs1, d["sigmas"][:n1], d["origins"][:1], nv, p["voxelsize"])
        kw1 = dict(box=None if d["box"] is None else d["box"][:1], max_images=sv.max_images, out=o1, ctx=ctx)
        for _ in range(250):                             # (the first small call of a context also builds its workspace;
            batch.voxelize_lattice_torch(*args1, **kw1)  #  ~10 ms of calls: the steady state of a screening loop -- the
                                                         #  per-call time still falls over the first few hundred calls)
        torch.cuda.synchronize(dev)
        runs = []
        for _ in range(5):                               # five runs of 40 calls back to back; the median is reported
            s0 = time.perf_counter()
            for _ in range(40):
                batch.voxelize_lattice_torch(*args1, **kw1)
            torch.cuda.synchronize(dev)
            runs.append((time.perf_counter() - s0) / 40 * 1e6)
        return sorted(runs)[2], [round(r, 2) for r in runs]

    # the one-molecule-per-call probe comes FIRST, on a quiet GPU as such a caller finds it (the same probe right after the
    # batch steps reads ~2 us more: clocks under load -- reported next to it)
    single0 = single_probe() if want_single else None
    # set-up, not a step: size both workspace sets of the context (device allocations happen on the
    # first call that uses a set) so that even --warmup 0 times no hipMalloc
    for _ in range(2):
        step()
    ctx.synchronize()
    # set-up too: let the clocks settle under this workload before anything is counted.  A GPU that has been idle (loading, host-side
    # workload generation) takes its first tens of milliseconds of kernels at lower clocks -- the 20 timed steps of the default run
    # are 43 ms: they read 2.158 ms per step where 6 s of the same steps read 2.105 (profiles/r5_bench_cfg2.json, `sustained`), and the
    # few-millisecond distance legs read 20 % low (bench_distances).  Untimed, reported as `clock_settle_s`; the W warm-up steps follow.
    settle = float(getattr(args, "settle_seconds", 0.0) or 0.0) if compute is None else 0.0
    if settle > 0:
        t_end = time.perf_counter() + settle
        while time.perf_counter() < t_end:
            for _ in range(8):
                step()
            ctx.synchronize()
    for _ in range(warmup):
        step()
    ctx.synchronize()                     # also surfaces asynchronous errors of the warm-up
    ctx.enable_kernel_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if os.environ.get("MKAMD_BENCH_KILL_RANK") == str(rank):      # tests/test_bench_launch.py: a rank that dies in the timed region
        os._exit(17)
    ctx.synchronize()
    mark_done(rank)                                               # this rank's steps are done (rank 0 counts these if a fence fails)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.enable_kernel_timing(False)
    k_ms, k_n = ctx.read_kernel_timing()
    kernel_name = ctx.last_tile_kernel() if hasattr(ctx, "last_tile_kernel") else None
    ctx.synchronize()
    elapsed = _max_over_ranks(elapsed, world)
    # sanity of what was produced inside the timed region (never a cached / skipped result)
    chk = out[0].double().sum().item()
    assert os.environ.get("MKAMD_DIAG") == "1" or (np.isfinite(chk) and chk > 0), "bench produced an empty grid"
    res = dict(p=p, nv=nv, V=V, C=C, elapsed=elapsed, k_ms=k_ms, k_n=k_n, alg=algorithmic_bytes(p, nv, C), kernel=kernel_name,
               topology=getattr(sv, "_topo", None) is not None)
    if "out" in keep:
        res["out"] = out

    if want_single:
        res["single_us_after_load"] = single_probe()[0]   # the same probe again, right after the batch steps (clocks under load)
        res["single_us"], res["single_us_runs"] = single0
    min_s = float(getattr(args, "min_seconds", 0.0) or 0.0)
    if sustain and min_s > 0 and elapsed > 0:
        # the same steps again, long enough to be seen from outside (same fences, max over ranks): every rank runs the
        # same number of steps, fixed up front from rank-independent numbers
        n2 = int(min(max(steps, np.ceil(min_s / (elapsed / steps))), 200000))
        # ... and, beside them, the shader clock the device sustains under THIS load: one wave on a stream of its own counts shader
        # clock ticks against the fixed 100 MHz reference for half a second (mkamd_clock_probe_dev).  Boxes differ here -- the tile
        # kernel takes the same 4.50 M cycles on every one of them, and four of them differed by 7 % in step time (docs/EXPERIMENTS_r5.md).
        ticks = probe_stream = None
        if compute is None and hasattr(ctx, "clock_probe_dev"):
            import torch
            probe_stream = torch.cuda.Stream(dev)
            ticks = torch.zeros(2, dtype=torch.int64, device=dev)
        fence()
        t0 = time.perf_counter()
        for i in range(n2):
            step()
            if ticks is not None and i == n2 // 4:
                try:
                    ctx.clock_probe_dev(probe_stream.cuda_stream, 500000, ticks.data_ptr())
                except Exception:                 # noqa: BLE001  (a reported extra)
                    ticks = None
        fence()
        e2 = _max_over_ranks(time.perf_counter() - t0, world)
        res["sustained"] = {"steps": n2, "seconds": round(e2, 4), "ms_per_step": round(e2 / n2 * 1e3, 4),
                            "value": round(world * B * V * C * n2 / e2 / 1e6, 2), "unit": "Mvoxel-channels/s"}
        if ticks is not None:
            probe_stream.synchronize()
            t_sh, t_ref = (int(v) for v in ticks.cpu().tolist())
            if t_ref > 0:
                res["sustained"]["shader_clock_ghz_under_load"] = round(t_sh / t_ref * 0.1, 4)
                res["sustained"]["shader_clock_source"] = "s_memtime ticks / s_memrealtime ticks (100 MHz) of one wave spinning 0.5 s beside the steps"

    def gather_legs():
        # (a secondary measurement: a failure in it -- RCCL, memory for the world x B result -- is reported on the line,
        #  it must not cost the primary one)
        if _RANKS["broken"]:
            res["gather_error"] = "skipped: a rank had already failed"
            return res
        try:
            # the trivial gather of the feature tensors, timed on its own: (a) one padded all-gather after the compute,
            # (b) chunk-overlapped with the compute (what a consumer that needs everything everywhere would run)
            fence()
            g0 = time.perf_counter()
            full = sv.gather(out)                 # (the first collective of a process also builds the communicator)
            fence()
            g0 = time.perf_counter()
            full = sv.gather(out)
            fence()
            res["gather_ms"] = (time.perf_counter() - g0) * 1e3
            assert full.shape[0] == world * B
            if "full_plain" in keep:
                res["full_plain"] = full
            del full
            full = sv.voxelize_gather(nchunks=4, loopback=world == 1 and compute is None)  # (the first point-to-point exchange sets its channels up)
            del full
            fence()
            g0 = time.perf_counter()
            full = sv.voxelize_gather(nchunks=4, loopback=world == 1 and compute is None)
            fence()
            res["gather_exchange"] = getattr(sv, "last_exchange", "?") + (" (one rank: batched send / receive to itself)" if world == 1 else "")
            both = (time.perf_counter() - g0) * 1e3
            res["compute_plus_overlapped_gather_ms"] = both
            res["gather_overlapped_extra_ms"] = both - elapsed / steps * 1e3
            assert full.shape[0] == world * B and torch.equal(full[rank * B:(rank + 1) * B], out)
            if "full_overlapped" in keep:
                res["full_overlapped"] = full
            del full
        except Exception as e:                                   # noqa: BLE001 -- reported, not swallowed
            res["gather_error"] = f"{type(e).__name__}: {e}"[:300]
            res.pop("gather_ms", None); res.pop("gather_overlapped_extra_ms", None)
            try:
                fence()
            except Exception:                                    # noqa: BLE001
                pass

        return res

    def release():
        nonlocal out, sv
        out = sv = None
        if dev.type == "cuda":
            torch.cuda.empty_cache()

    if want_gather and defer_gather:
        res["_gather_legs"], res["_release"] = gather_legs, release     # main() runs them last: they create the RCCL communicator
        return res
    if want_gather:
        gather_legs()
    release()
    return res


_FLOPS_MEMO = {}


def roofline_of(res, workload, B, tile_k):
    k_avg_ms = res["k_ms"] / max(res["k_n"], 1)
    achieved = res["alg"] / (k_avg_ms * 1e-3) / 1e9 if res["k_n"] else None
    kernel = res.get("kernel") or None
    entry, src, step_traffic = pmc_entry(workload, B, tile_k, kernel)
    traffic = int((entry["WRITE_SIZE"] + 2.0 * entry["FETCH_SIZE"]) * 1024) if entry else None
    out = {"bound": "hbm", "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
           "traffic": traffic, "traffic_of": "the dominant kernel alone (per launch)", "step_traffic": step_traffic,
           "traffic_source": src, "kernel": kernel, "timed_region": "the tile kernel + k_tail (what runs between the library's timing events)",
           "kernel_avg_ms": round(k_avg_ms, 5), "kernel_launches": int(res["k_n"]),
           "algorithmic_bytes_per_launch": int(res["alg"])}
    # the secondary roofline SURVEY.md section 8d / 7-H3 asks for next to the HBM one: vector FP32
    try:
        if (workload, B) not in _FLOPS_MEMO:
            _FLOPS_MEMO[(workload, B)] = algorithmic_flops(res["p"], res["nv"], res["C"])
        flops, pairs_per_voxel = _FLOPS_MEMO[(workload, B)]
        tf = flops / (k_avg_ms * 1e-3) / 1e12 if res["k_n"] else None
        sec = {"bound": "valu", "achieved": round(tf, 2) if tf else None, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": round(tf / FP32_VALU_PEAK_TFLOPS, 4) if tf else None, "algorithmic_flops_per_launch": flops,
               "flop_per_byte": round(flops / res["alg"], 2), "in_range_pairs_per_voxel": round(pairs_per_voxel, 2),
               "formula": "22 x in-range (voxel, atom x channel) pairs + 14 x voxel-channels (SURVEY.md 8d)"}
        if entry and entry.get("SQ_WAVES") and entry.get("SQ_INSTS_VALU") is not None:
            sec["valu_insts_per_wave"] = round(entry["SQ_INSTS_VALU"] / entry["SQ_WAVES"], 1)
            if kernel and "k_voxelize_tiles" in kernel and "team" not in kernel:
                sec["valu_insts_per_tile"] = sec["valu_insts_per_wave"]        # one wave per 512-voxel tile
            # shader cycles of the launch: GRBM_GUI_ACTIVE is summed over the chip's 8 XCDs, SQ_BUSY_CYCLES over its 32 shader engines
            clk = (entry["GRBM_GUI_ACTIVE"] / 8.0) if entry.get("GRBM_GUI_ACTIVE") else (entry.get("SQ_BUSY_CYCLES", 0) / 32.0)
            if clk and entry.get("SQ_ACTIVE_INST_VALU") is not None:
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1 024 SIMDs
                sec["valu_busy"] = round(entry["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / clk, 3)
                sec["valu_busy_of"] = "SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / " + ("(GRBM_GUI_ACTIVE / 8 XCDs)" if entry.get("GRBM_GUI_ACTIVE") else "(SQ_BUSY_CYCLES / 32 shader engines)")
            sec["counters_source"] = src
        out["secondary"] = sec
    except Exception as e:                             # noqa: BLE001 -- a secondary number
        out["secondary"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(DEFAULT_BATCH))
    ap.add_argument("--batch", type=int, default=0, help="items per GPU per step (0 = workload default)")
    ap.add_argument("--tile-k", type=int, default=0)
    ap.add_argument("--lds-tier", type=int, default=-1, help="-1 adaptive (default), 0/1/2 = 640/768/1024 LDS entries per tile")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather-timeout", type=float, default=100.0,
                    help="seconds the (untimed, secondary) RCCL feature-gather legs may take before they are abandoned (below the process group's own 120 s, whose expiry ends the process without the line)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary workloads (cfg1/cfg3/cfg4/cfg5 at N=1, the cfg3 batched-molecule leg at N>1)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap step n+1's binning pre-pass with step n's tile kernel")
    ap.add_argument("--min-seconds", type=float, default=6.0,
                    help="after the timed K steps, keep stepping until this much wall time has been spent on the same workload and "
                         "report it as `sustained` (the K-step region of a 64^3 workload is tens of milliseconds: too short for a "
                         "utilisation sampler to see, and for the clocks to settle). 0 = skip")
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed steps of the workload before the warm-up steps, so that the timed steps run at settled clocks (0 = none)")
    ap.add_argument("--no-topology", action="store_true",
                    help="cfg1 / cfg4 (rotated copies / frames of one molecule): the plain call on the sigma matrix repeated per frame instead of the topology handle (A-B)")
    ap.add_argument("--no-single", action="store_true", help="skip the single-grid latency probe (profiling passes: every launch is a full batch)")
    ap.add_argument("--value-tol", type=float, default=0.0,
                    help="opt into the tolerance-aware reach (mkamd_ctx_set_value_tolerance): atoms are culled where they are "
                         "worth less than this (<= 1e-5). 0 (default) = the reference's hard 5 A cutoff; reported in `config`")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only: rendezvous (gloo), shard and gather a tiny batch with a stand-in compute; no timing")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under torchrun: become the launcher (one rank per GPU, RCCL), the line comes from rank 0 of the children
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    _json_only_stdout()
    if args.dry_run:
        return bench_distances_sharded(args, dry=True) if args.workload == "dist" else dry_run(args)
    if args.workload == "dropin":
        return bench_dropin(args)
    if args.workload == "dist":
        # one GPU, not under torchrun: the leg-by-leg distance line; N ranks (or one under torchrun): the frames-sharded path
        if args.gpus != 1 or "RANK" in os.environ:
            return bench_distances_sharded(args)
        return bench_distances(args)

    import torch
    import torch.distributed as dist

    from moleculekit_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    # REHEARSAL of the multi-process path on a box with fewer GPUs than ranks (MKAMD_BENCH_SHARE_DEVICES=1): ranks share devices
    # round-robin.  Everything between the processes is real -- torchrun, the rendezvous, the gloo fences, the max over ranks,
    # RCCL's answer to two ranks on one device in the gather legs -- the numbers are not a scaling measurement and the line says so.
    shared_devices = os.environ.get("MKAMD_BENCH_SHARE_DEVICES", "0") == "1" and torch.cuda.device_count() > 0
    if shared_devices:
        local = local % torch.cuda.device_count()
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun: RCCL even for one rank (exercises the path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # gloo for the fences / the max over ranks (CPU tensors), RCCL for the feature gathers: the RCCL communicator is
        # created by the first GPU collective, i.e. by the gather legs at the very end -- its mere existence slows every
        # kernel of the process by 3-7 % (_max_over_ranks), and the timed region has no collective to need it
        import datetime
        # (a collective that cannot complete raises after two minutes instead of holding the node: the gather legs are
        #  secondary measurements wrapped in try / except, the headline line still prints)
        dist.init_process_group("cpu:gloo,cuda:nccl", timeout=datetime.timedelta(seconds=120))

    B = args.batch or DEFAULT_BATCH[args.workload]
    ctx = _lib.default_context(local)
    ctx.set_tile_k(args.tile_k)
    ctx.set_lds_tier(args.lds_tier)
    ctx.set_prepass_mode(int(os.environ.get("MKAMD_PREPASS", "-1")))
    ctx.set_force_general(os.environ.get("MKAMD_FORCE_GENERAL", "0") == "1")
    ctx.set_fine_cells(os.environ.get("MKAMD_FINE_CELLS", "0") == "1")         # A-B knob: half-cutoff cells
    ctx.set_tile_items(int(os.environ.get("MKAMD_TILE_ITEMS", "-1")))          # A-B knob: a workgroup per item (ligand-sized batches)
    # steps are independent batches whose inputs are resident before the loop: ShardedVoxelizer promises them to the
    # library per call, which then overlaps the pre-pass of step n+1 with the tile kernel of step n -- the package's
    # own product path (run_workload), not a knob of this file
    ctx.set_value_tolerance(args.value_tol)
    ctx.set_direct_binning(int(os.environ.get("MKAMD_DIRECT", "-1")))          # A-B knob: one-pass direct binning (-1 auto, 0 off, 1 on)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            _collective(lambda: dist.all_reduce(torch.zeros(1)))      # the barrier, on a CPU tensor: the gloo side of the group
        torch.cuda.synchronize(dev)

    if use_dist and rank == 0:
        _RANKS["emit"] = lambda: print(json.dumps({"metric": "Mvoxel-channels/s (64^3 grid, 8 ch)", "value": None, "unit": "Mvoxel-channels/s",
                                                    "n_gpus": world, "ranks_alive": ranks_done(world), "degraded": _RANKS["broken"],
                                                    "error": "the job was taken down before rank 0 finished its timed region"}), flush=True)
        term_reporter()

    def dropin_probe():
        """The reference's own call pattern: ONE molecule per synchronous call, host arrays in, float64 [V, C] out
        (BASELINE.json configs[0]: 3PTB, 24^3 @ 1 A) -- what a user who only swaps the import sees.  -> (ms per call, max |err|)"""
        from moleculekit_amd.voxeldescriptors import getVoxelDescriptors
        g3 = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_3ptb.npz"))
        kw = dict(boxsize=[24, 24, 24], center=g3["center"], voxelsize=1, usercoords=g3["coords"], userchannels=g3["sigmas"])
        for _ in range(10):
            f3, _, _ = getVoxelDescriptors(None, **kw)
        runs = []
        for _ in range(3):                                    # three runs of 100 calls, the median is reported
            t0 = time.perf_counter()
            for _ in range(100):
                f3, _, _ = getVoxelDescriptors(None, **kw)
            runs.append((time.perf_counter() - t0) / 100 * 1e3)
        return sorted(runs)[1], float(np.abs(f3 - g3["features"]).max())

    # like the single-grid probe inside run_workload: first, on a quiet GPU, and once more after the batch legs
    dropin0 = None
    if world == 1 and rank == 0 and not args.no_extra and args.workload == "cfg2":
        try:
            dropin0 = dropin_probe()
        except Exception as e:                 # noqa: BLE001 -- a secondary number
            dropin0 = f"{type(e).__name__}: {e}"[:300]
    res = run_workload(args.workload, B, args.steps, args.warmup, ctx, dev, rank, world, args, fence,
                       want_gather=use_dist and not args.no_gather, want_single=rank == 0 and not args.no_single, sustain=True,
                       defer_gather=True)
    p, nv, V, C, elapsed = res["p"], res["nv"], res["V"], res["C"], res["elapsed"]

    # secondary legs (every rank takes part; only rank 0 reports).  N = 1: the other BASELINE configs, so that their
    # roofline fractions are timed by whoever runs this file; N > 1: the batched-molecule workload (cfg3) through the
    # same sharded path -- north_star's "batched molecules" -- next to the headline 64^3 workload.
    extra = {}
    if not args.no_extra and args.workload == "cfg2" and not args.batch:
        names = ["cfg1", "cfg3", "cfg4", "cfg5"] if world == 1 else ["cfg3"]
        for nm in names:
            ctx.set_lds_tier(args.lds_tier)
            try:                                   # a secondary leg that fails is reported, the headline line still prints
                # (the same steps and warm-up as the headline: with a quarter of the steps after two warm-up calls the 3PTB
                #  batch read 2.0 ms per step against 1.85 sustained -- the pipeline's fill and the clocks after seconds of
                #  host-side workload generation)
                r2 = run_workload(nm, DEFAULT_BATCH[nm], max(3, args.steps), max(2, args.warmup), ctx, dev, rank, world, args, fence)
            except Exception as e:                 # noqa: BLE001
                extra[nm] = {"error": f"{type(e).__name__}: {e}"[:300]}
                continue
            steps2 = max(3, args.steps)
            extra[nm] = {"value": round(world * DEFAULT_BATCH[nm] * r2["V"] * r2["C"] * steps2 / r2["elapsed"] / 1e6, 2),
                         "unit": "Mvoxel-channels/s", "items_per_gpu_per_step": DEFAULT_BATCH[nm], "steps": steps2,
                         "ms_per_step": round(r2["elapsed"] / steps2 * 1e3, 4), "grid": [int(v) for v in r2["nv"]],
                         **({"topology_reuse": "frames of one molecule: sigma classes / class ids built once (mkamd_topology), not per call"} if r2.get("topology") else {}),
                         "roofline": roofline_of(r2, nm, DEFAULT_BATCH[nm], args.tile_k)}
            if r2.get("topology"):
                # ... and the same leg WITHOUT the handle (the plain call, class discovery inside the timed region: what rounds 1-4 timed),
                # so that the two are never mistaken for each other
                try:
                    plain_args = argparse.Namespace(**{**vars(args), "no_topology": True})
                    r2p = run_workload(nm, DEFAULT_BATCH[nm], steps2, max(2, args.warmup), ctx, dev, rank, world, plain_args, fence)
                    extra[nm]["plain_call"] = {"ms_per_step": round(r2p["elapsed"] / steps2 * 1e3, 4),
                                               "value": round(world * DEFAULT_BATCH[nm] * r2p["V"] * r2p["C"] * steps2 / r2p["elapsed"] / 1e6, 2)}
                except Exception as e:             # noqa: BLE001
                    extra[nm]["plain_call"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    if not args.no_extra and args.workload == "cfg2" and not args.batch and world == 1 and args.value_tol == 0.0:
        # the opt-in tolerance-aware reach (mkamd_ctx_set_value_tolerance, eps = 1e-6) on the headline workload: atoms
        # are culled per tile where they are worth less than eps -- a secondary number, the headline keeps the hard cutoff
        try:
            ctx.set_value_tolerance(1e-6)
            ctx.set_lds_tier(args.lds_tier)        # (the adaptive tier restarts from this workload's own statistics)
            steps2 = max(3, args.steps // 2)
            r3 = run_workload("cfg2", B, steps2, 4, ctx, dev, rank, world, args, fence)
            extra["cfg2_value_tolerance_1e-6"] = {
                "value": round(world * B * r3["V"] * r3["C"] * steps2 / r3["elapsed"] / 1e6, 2), "unit": "Mvoxel-channels/s",
                "items_per_gpu_per_step": B, "steps": steps2, "ms_per_step": round(r3["elapsed"] / steps2 * 1e3, 4),
                "note": "opt-in: values within 1e-6 of the exact mode (tests/test_gpu_parity.py), not bit-identical",
                "roofline": {k: v for k, v in roofline_of(r3, "cfg2", B, args.tile_k).items() if k not in ("traffic", "step_traffic", "traffic_source", "traffic_of", "secondary")}}
        except Exception as e:                     # noqa: BLE001
            extra["cfg2_value_tolerance_1e-6"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            ctx.set_value_tolerance(0.0)

    if not args.no_extra and args.workload == "cfg2" and not args.batch and world == 1:
        # the package's streaming drivers on cfg4-shaped work: pipelined by promise (device-resident source), and fed from an XTC file
        raw4 = extra.get("cfg4", {}).get("ms_per_step")
        for nm, fn in (("stream_cfg4", bench_stream_cfg4), ("xtc_cfg4", bench_xtc_cfg4)):
            try:
                extra[nm] = fn(ctx, dev, raw4)
            except Exception as e:                 # noqa: BLE001
                extra[nm] = {"error": f"{type(e).__name__}: {e}"[:300]}

    printed = threading.Lock()
    timed_out = []                                 # non-empty once the watchdog of the gather legs has fired

    def emit():
        """rank 0's ONE JSON line (called once: after the gather legs, or by their watchdog)"""
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            alive = world if not _RANKS["broken"] else max(ranks_done(world), 1)
            total_vc = alive * B * V * C * args.steps
            k_avg_ms = res["k_ms"] / max(res["k_n"], 1)
            info = ctx.device_info()
            line = {
                "metric": "Mvoxel-channels/s (64^3 grid, 8 ch)" if args.workload == "cfg2" else f"Mvoxel-channels/s ({args.workload})",
                "value": round(total_vc / elapsed / 1e6, 2),
                "unit": "Mvoxel-channels/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.workload}: BASELINE.json configs[{int(args.workload[3:]) - 1}]",
                           "items_per_gpu_per_step": B, "grid": [int(v) for v in nv], "channels": C,
                           "voxelsize": p["voxelsize"], "atoms_per_gpu": int(p["atom_offsets"][-1]),
                           "periodic": p["box"] is not None, "tile_k": args.tile_k, "pipelined_steps": not args.no_pipeline,
                           "value_tolerance": args.value_tol, "topology_reuse": bool(res.get("topology")), "clock_settle_s": args.settle_seconds,
                           **({"rehearsal": "ranks SHARE devices (MKAMD_BENCH_SHARE_DEVICES=1): the multi-process path on fewer GPUs than ranks, not a scaling measurement"} if shared_devices else {}),
                           "parallelism": f"dp{world} (items sharded: every rank loads, stages and keeps only its own shard; no collective in the timed region; "
                                          "fences over gloo, feature gathers over RCCL after everything timed)",
                           "device": info["name"], "arch": info["arch"], "compute_units": info["compute_units"]},
                "roofline": roofline_of(res, args.workload, B, args.tile_k),
                "tile_kernel_share_of_step": round(k_avg_ms / (elapsed / args.steps * 1e3), 4) if res["k_n"] else None,
                "sustained": res.get("sustained"),
                "single_grid_latency_us": round(res["single_us"], 2) if "single_us" in res else None,
                "single_grid_latency_us_runs": res.get("single_us_runs"),
                "single_grid_latency_after_batch_load_us": round(res["single_us_after_load"], 2) if "single_us_after_load" in res else None,
                "gather_ms": round(res["gather_ms"], 3) if "gather_ms" in res else None,
                "gather_overlapped_extra_ms": round(res["gather_overlapped_extra_ms"], 3) if "gather_overlapped_extra_ms" in res else None,
                **({"gather_error": res["gather_error"]} if "gather_error" in res else {}),
                **({"gather_exchange": res["gather_exchange"]} if "gather_exchange" in res else {}),
                "ranks_alive": alive,
                **({"degraded": f"a collective between the ranks failed ({_RANKS['broken']}): `value` counts the {alive} rank(s) that finished "
                                "the timed region, timed on rank 0 alone"} if _RANKS["broken"] else {}),
            }
            if extra:
                line["other_workloads" if world == 1 else "batched_molecules"] = extra
            if world == 1 and not args.no_extra and args.workload == "cfg2" and not timed_out:     # (no more GPU work behind a hung leg)
                try:
                    if isinstance(dropin0, tuple):
                        line["dropin_call_ms"] = round(dropin0[0], 4)
                        line["dropin_max_abs_err_vs_reference"] = dropin0[1]
                        line["dropin_call_after_batch_load_ms"] = round(dropin_probe()[0], 4)
                    elif dropin0 is not None:
                        line["secondary_error"] = dropin0
                    # the distance_utils row next to it: dist_trajectory, bit-exact float32
                    dargs = argparse.Namespace(batch=0, steps=max(3, args.steps // 4), warmup=2, no_cpu_baseline=True)
                    dl = bench_distances(dargs, emit=False)
                    line.setdefault("other_workloads", {})["dist_trajectory"] = {
                        "value": dl["value"], "unit": dl["unit"], "ms_per_step": dl["ms_per_step"], "config": dl["config"]["workload"],
                        "roofline": dl["roofline"], "nonperiodic": dl["nonperiodic"],
                        **({"selfdist": dl["selfdist"], "small_call": dl["small_call"]} if "selfdist" in dl else {})}
                except Exception as e:             # noqa: BLE001 -- secondary numbers: reported, the headline line still prints
                    line["secondary_error"] = f"{type(e).__name__}: {e}"[:300]

            if world == 1 and not args.no_cpu_baseline and not timed_out:
                line["cpu_baseline"] = cpu_baseline(args.workload)
            print(json.dumps(line), flush=True)

    _RANKS["emit"] = emit                          # from here on a SIGTERM prints the real line (rank 0; emit() prints once)
    if "_gather_legs" in res:                      # last: the first RCCL collective of the process
        # Nothing with N > 1 has ever run on RCCL where this file was written: should the legs hang (they come after
        # everything timed), the headline line must still come out -- a watchdog in every rank reports the time-out on
        # the line (rank 0) and ends the process.
        legs, rel = res.pop("_gather_legs"), res.pop("_release")

        def abandon():
            timed_out.append(True)
            res["gather_error"] = f"the feature-gather legs did not return within {args.gather_timeout:.0f} s (abandoned; everything timed was done before them)"
            for k in ("gather_ms", "gather_overlapped_extra_ms", "compute_plus_overlapped_gather_ms"):
                res.pop(k, None)
            try:
                emit()
            finally:
                os._exit(0)

        guarded(lambda: (legs(), rel()), args.gather_timeout, abandon)

    emit()

    if use_dist:
        _collective(lambda: dist.all_reduce(torch.zeros(1)))
        if _RANKS["broken"]:
            os._exit(1 if rank else 0)             # (no orderly shutdown with a dead peer: rank 0 has printed its line)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
