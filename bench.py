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

# the parts (round 6: this file was a 1 700-line monolith; the contract, the command line, run_workload and main stay here)
from tools.benchlib.workloads import (DEFAULT_BATCH, FP32_VALU_PEAK_TFLOPS, HBM_PEAK_GBS, algorithmic_bytes, algorithmic_flops,  # noqa: E402,F401
                                      in_range_pairs, make_config, make_workload, real_protein_config, reduction_workload)
from tools.benchlib.evidence import dist_traffic, pmc_entry, reduction_pmc, roofline_of  # noqa: E402,F401
from tools.benchlib.baseline import cpu_baseline, host_cores_granted, host_cpu_model  # noqa: E402,F401
from tools.benchlib.launch import (_RANKS, _collective, _free_port, _json_only_stdout, _max_over_ranks, _store, guarded, launch_ranks,  # noqa: E402,F401
                                   mark_done, ranks_done, term_reporter)
from tools.benchlib.distances import bench_contacts, bench_distances, bench_distances_sharded, bench_reductions  # noqa: E402,F401
from tools.benchlib.streams import bench_dropin, bench_stream_cfg4, bench_xtc_cfg4  # noqa: E402,F401


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
                          "collectives_exercised": res.get("collectives_exercised"),
                          "seconds": round(time.perf_counter() - t0, 3)}), flush=True)
    dist.destroy_process_group()
    if not all(flags):
        raise SystemExit("dry run: sharding / gather mismatch")


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
            # every other collective an N-rank run can issue, once each, checked against the local rows (with one
            # rank under torchrun each of them runs on RCCL at least once -- all_gather_into_tensor above, gather-to-root and the chunked
            # forms of both exchanges here)
            paths = {"all_gather_into_tensor (gather)": True, "p2p nchunks=4": True}
            loop = world == 1 and compute is None
            mine = slice(rank * B, (rank + 1) * B)
            root = sv.gather(out, dst=0)
            paths["gather-to-root (padded dist.gather)"] = (root is None) if rank else bool(torch.equal(root[mine], out))
            del root
            for nch in (1, 4):
                full = sv.voxelize_gather(nchunks=nch, exchange="p2p", loopback=loop)
                paths[f"p2p nchunks={nch}"] = bool(torch.equal(full[mine], out))
                full = sv.voxelize_gather(nchunks=nch, exchange="allgather")
                paths[f"chunked all_gather_into_tensor nchunks={nch}"] = bool(torch.equal(full[mine], out))
                full = sv.voxelize_gather(nchunks=nch, dst=0)
                paths[f"chunked gather-to-root nchunks={nch}"] = (full is None) if rank else bool(torch.equal(full[mine], out))
                del full
            fence()
            res["collectives_exercised"] = paths
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
                **({"collectives_exercised": res["collectives_exercised"]} if "collectives_exercised" in res else {}),
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
