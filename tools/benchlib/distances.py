"""`--workload dist`: the distance_utils row -- dist_trajectory legs, group reductions, contact lists on one GPU;
frames sharded over the ranks for N > 1 (and the CPU dry run of that path)."""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np

from .evidence import dist_traffic, reduction_pmc
from .launch import _RANKS, _collective, _max_over_ranks, guarded, mark_done, ranks_done, term_reporter
from .workloads import DEFAULT_BATCH, HBM_PEAK_GBS, ROOT, reduction_workload

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
        line["cdist_pdist"] = bench_cdist_pdist(args, ctx, dev, busy, check)
        line["reduction"] = bench_reductions(args, ctx, dev, busy, check)
        line["contacts"] = bench_contacts(args, ctx, dev, busy, check, coords, box, chains, chains_h, s1, s2, d1, d2, F)
        line["host_call"] = bench_host_call(coords, box, chains_h, s1, s2, F)
    del coords, out
    torch.cuda.empty_cache()
    if emit:
        print(json.dumps(line), flush=True)
    return line


def bench_host_call(coords, box, chains_h, s1, s2, F):
    """The PCIe-inclusive figure of the f-1 row (never `value`): the reference-shaped call itself -- moleculekit_amd.distance_utils.dist_trajectory
    with HOST arrays in and out, what `install()` puts under MetricDistance -- for the headline call (819 MB of distances down) and for
    MetricDistance's usual small one (300 x 30: 74 MB down).  Of the 737-MB coordinate array only the selected atoms' rows go up
    (csrc/host_pack.h: 700 and 330 of 30 000 atoms)."""
    from moleculekit_amd import distance_utils as du
    ch, bh = coords.cpu().numpy(), box.cpu().numpy()
    out = {}
    from moleculekit_amd import _lib
    ctx = _lib.default_context(0)
    for name, a, b in (("200 x 500", s1, s2), ("300 x 30", s2[:300].copy(), s1[:30].copy())):
        res = np.empty((F, len(a) * len(b)), np.float32)
        ms = {}
        for label, mask in (("packed", 0), ("whole_array", 32)):       # same-box A-B: bit 32 = upload the whole trajectory (before round 6's host_pack.h)
            ctx.set_dist_kernels(mask)
            du.dist_trajectory(ch, bh, a, b, chains_h, False, True, res)
            t0 = time.perf_counter()
            for _ in range(2):
                du.dist_trajectory(ch, bh, a, b, chains_h, False, True, res)
            ms[label] = (time.perf_counter() - t0) / 2
        ctx.set_dist_kernels(0)
        dt = ms["packed"]
        out[name] = {"ms_per_call": round(dt * 1e3, 2), "ms_per_call_whole_array_uploaded": round(ms["whole_array"] * 1e3, 2),
                     "Mdist_per_s": round(F * len(a) * len(b) / dt / 1e6, 1),
                     "coords_array_MB": round(ch.nbytes / 1e6, 1),
                     "bytes_over_pcie_MB": round((len(np.union1d(a, b)) * 3 * F * 4 + res.nbytes) / 1e6, 1)}
    out["what"] = ("host numpy arrays in and out (pageable memory), periodic by chain; only the selected atoms' rows of the trajectory "
                   "are uploaded (host_pack.h)")
    return out


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
    res["one_frame"] = bench_reduction_one_frame(args, ctx, dev, busy, check)
    pmc = reduction_pmc()
    if pmc:
        res["periodic"]["valu"].update(pmc)
    return res


def bench_reduction_one_frame(args, ctx, dev, busy, check):
    """The residue-contact map of ONE structure (200 residues of 15 atoms, all 19 900 pairs, 4.5 M atom pairs; periodic): calls of few
    frames take k_dist_reduction_few (lanes along the second groups) -- timed beside the kernel whose lanes are frames (one lane in 64
    at work), the whole result checked against the oracle bit for bit."""
    import torch
    G, A = 200, 15
    out = {}
    for F in (1, 8):
        coords, box, atoms, offs, chains, masses = reduction_workload(G, A, F)
        N = coords.shape[0]
        t = lambda a: torch.as_tensor(a, device=dev)
        d_c, d_b, d_a, d_o, d_m, d_ch = t(coords), t(box), t(atoms), t(offs), t(masses), t(chains.astype(np.int32))
        P = G * (G - 1) // 2
        o = torch.empty((F, P), device=dev, dtype=torch.float32)
        call = lambda: ctx.dist_reduction_dev(d_c, N, F, d_b, d_a, d_o, G, N, d_a, d_o, G, d_ch, d_ch, True, False, True, d_m, 0, 0, o)
        entry = {}
        for name, block in (("us_per_call", 0), ("frame_lane_kernel_us", 8)):
            ctx.set_reduction_block(block)
            busy(call, 0.15)
            if check and block == 0:
                from oracle import oracle
                groups = [atoms[offs[g]:offs[g + 1]].tolist() for g in range(G)]
                ref = oracle.dist_trajectory_reduction(coords, box, groups, groups, chains, chains, True, True, masses, 0, 0)
                if not np.array_equal(o.cpu().numpy(), ref, equal_nan=True):
                    raise SystemExit(f"dist_trajectory_reduction of {F} frame(s) on the GPU is not bit-exact with the oracle")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                call()
            e1.record()
            torch.cuda.synchronize(dev)
            entry[name] = round(e0.elapsed_time(e1) / 50 * 1e3, 1)
        ctx.set_reduction_block(0)
        out[f"{F}_frame" + ("s" if F > 1 else "")] = entry
    out["kernel"] = "mkamd::k_dist_reduction_few"
    return out


def bench_cdist_pdist(args, ctx, dev, busy, check):
    """cdist / pdist (distance_utils.pyx:355-416) on device pointers: 8 192 x 8 192 points in 3-D (268 MB of result) and the condensed
    upper triangle of 16 384 points (537 MB) -- store-bound: algorithmic bytes = the result once + the points once."""
    import torch
    rng = np.random.default_rng(12)
    n1 = n2 = 8192
    a = rng.normal(0, 20, size=(n1, 3)).astype(np.float32)
    b = rng.normal(0, 20, size=(n2, 3)).astype(np.float32)
    np_ = 16384
    c = rng.normal(0, 20, size=(np_, 3)).astype(np.float32)
    d_a, d_b, d_c = (torch.as_tensor(x, device=dev) for x in (a, b, c))
    out_c = torch.empty((n1, n2), device=dev, dtype=torch.float32)
    out_p = torch.empty((np_ * (np_ - 1) // 2,), device=dev, dtype=torch.float32)
    res = {}
    for name, call, alg, out in (("cdist", lambda: ctx.cdist_dev(d_a, n1, d_b, n2, 3, out_c), n1 * n2 * 4 + (n1 + n2) * 12, out_c),
                                 ("pdist", lambda: ctx.pdist_dev(d_c, np_, 3, out_p), np_ * (np_ - 1) // 2 * 4 + np_ * 12, out_p)):
        busy(call, 0.2)
        for _ in range(max(3, args.warmup)):
            call()
        torch.cuda.synchronize(dev)
        if check:
            from oracle import oracle
            if name == "cdist":
                ok = np.array_equal(out[:64].cpu().numpy(), oracle.cdist(a[:64], b))
            else:
                ok = np.array_equal(out[:np_ - 1].cpu().numpy(), oracle.cdist(c[:1], c[1:])[0])      # row 0 of the triangle
            if not ok:
                raise SystemExit(f"{name} on the GPU is not bit-exact with the oracle")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            call()
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / args.steps
        res[name] = {"shape": f"{n1} x {n2} x 3" if name == "cdist" else f"{np_} points x 3", "us_per_call": round(ms * 1e3, 2),
                     "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                                  "algorithmic_bytes_per_launch": alg}, "kernel": "mkamd::k_" + name + "_rows<3>"}
    return res


def bench_contacts(args, ctx, dev, busy, check, coords, box, chains, chains_h, s1, s2, d1, d2, F):
    """contacts_trajectory (distance_utils.pyx:59-93) on device pointers: the dist leg's 200 x 500 pairs over its F frames,
    threshold 8 A, periodic by chain -- counted (every distance once, second atoms in registers: dist_kernels.h, the rectangular contact kernels;
    the contact masks are kept), scanned per group of rows and compacted on the device, the list stays in HBM.  Algorithmic bytes: the selected atoms' coordinates once + 8 B per contact + the frame offsets."""
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
    # get_collisions (distance_utils.pyx:98-121; molecule.py:3731: what Molecule.append(collisiondist=...) sends): ONE frame, no box, host arrays in and
    # the Python list out -- calls of few frames count with lanes along the second atoms (k_contacts_count_rect_few)
    from moleculekit_amd.distance_utils import get_collisions
    rng = np.random.default_rng(2)
    ca, cb = rng.uniform(0, 60, size=(20000, 3)).astype(np.float32), rng.uniform(0, 60, size=(3000, 3)).astype(np.float32)
    hits = get_collisions(ca, cb, 1.3)
    t0 = time.perf_counter()
    for _ in range(5):
        get_collisions(ca, cb, 1.3)
    coll_ms = (time.perf_counter() - t0) / 5 * 1e3
    collisions = {"shape": "20000 x 3000 atoms, one frame, 1.3 A", "ms_per_call": round(coll_ms, 3), "collisions": len(hits) // 2,
                  "pair_tests_per_s_G": round(20000 * 3000 / coll_ms / 1e6, 1), "what": "the whole Python call: host arrays in, list out"}
    return {"get_collisions": collisions,
            "shape": f"{n1} x {n2} pairs x {F} frames, threshold {thr} A, periodic: {n} contacts", "ms_per_call": round(ms, 4),
            "pair_tests_per_s_G": round(n1 * n2 * F / ms / 1e6, 1),
            "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 5),
                         "algorithmic_bytes_per_launch": alg, "note": "VALU-bound: k_contacts_count_rect spends 18.9 lane-instructions on a periodic pair (packed arithmetic, the image integers behind one accumulated test, 2.4 of them the contact bit) = 118 us of issue at 2.0 GHz for this shape, measured 127; the fill pass reads the 16-bit masks the count pass kept; wall clock incl. the count read-back"},
            "kernel": "mkamd::k_contacts_count_rect<true> + k_contacts_scan + k_contacts_fill_rect"}


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
            full = sd.gather(out)                  # (the first collective of a process also builds the communicator)
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
