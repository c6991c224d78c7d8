"""Secondary legs: the drop-in call itself, and cfg4 frames streamed from a resident trajectory / an XTC file."""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np

from .workloads import DEFAULT_BATCH, HBM_PEAK_GBS, ROOT, make_workload

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


def bench_xtc_cfg4(ctx, dev, raw_ms_per_step, frames=2048, chunk=256, frames_gpu=16384, chunk_gpu=2048, ramp_gpu=512):
    """Secondary leg `xtc_cfg4`: the cfg4 FEEDER -- a synthetic 30 000-atom XTC trajectory (64 frames of the cfg4 random walk
    written with moleculekit_amd.xtc.write_xtc, the records repeated: XTC frames are self-contained) voxelized through
    batch.iterVoxelizeXTC, with the coordinates decompressed ON THE DEVICE (decode="auto": csrc/xtc_gpu.h, large chunks) and,
    beside it, by libmkamd.so's host threads (decode="host", the round-3 path).  Reports the end-to-end rates, the host
    decoder's rate alone and how idle the GPU is (the voxelizer's share of the wall time at the raw cfg4 step).
    Round 6: the feed's copy/decode stream is a high-priority stream (batch._stream_voxelize: on a hardware queue shared with the voxelizer's
    streams the walk and the tile kernel were time-sliced and the leg flipped between two rates); 2 048 frames per chunk approached through 512 and
    1 024 is the best plan for a 16 384-frame job (194 k frames/s three times out of three; 175-192 k without the ramp), 4 096 the best steady rate
    (docs/EXPERIMENTS_r6.md section 12)."""
    import tempfile
    import torch
    from moleculekit_amd import _lib, batch, xtc
    if os.environ.get("MKAMD_BENCH_XTC_PLAN"):                        # "chunk,ramp": A-B of chunk plans inside the whole line (tools/gpu_session.sh)
        chunk_gpu, ramp_gpu = (int(v) for v in os.environ["MKAMD_BENCH_XTC_PLAN"].split(","))
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

        def run(decode, nframes, nchunk, ramp=0):
            n, marks = 0, []
            for idx, feats in batch.iterVoxelizeXTC(fn, sig, p["centers"][0], p["boxsize"], p["voxelsize"], pbc=True, chunk=nchunk, ctx=ctx,
                                                    frames=np.arange(nframes), decode=decode, ramp=ramp):
                n += len(idx)
                ev = torch.cuda.Event(enable_timing=True)                       # when this chunk's features are complete on the device
                ev.record(torch.cuda.current_stream(dev))
                marks.append((ev, n))
                del feats
            torch.cuda.synchronize(dev)
            return n, marks

        def leg(decode, nframes, nchunk, ramp=0):
            run(decode, nframes, nchunk, ramp)                                  # warm: buffers, pinned staging, the allocator's blocks
            t0 = time.perf_counter()
            n, marks = run(decode, nframes, nchunk, ramp)
            dt = time.perf_counter() - t0
            o = {"frames": n, "frames_per_call": nchunk, "frames_per_s": round(n / dt, 1), "Matoms_per_s": round(n * N / dt / 1e6, 1)}
            if ramp:
                o["first_call_frames"] = ramp
            if len(marks) >= 4:                                                 # the feed once it is full: from the first chunk of the FULL size
                full = [i for i in range(1, len(marks)) if marks[i][1] - marks[i - 1][1] == nchunk]
                i0 = full[0] if ramp and len(full) >= 2 else 1                  # (no ramp: chunk 2's features complete -> the last chunk's)
                (ea, na), (eb, nb) = marks[i0], marks[full[-1] if ramp and len(full) >= 2 else -1]
                if nb > na:
                    o["steady_frames_per_s"] = round((nb - na) / (ea.elapsed_time(eb) * 1e-3), 1)
            if per_frame_s:
                o["gpu_busy_fraction"] = round(n * per_frame_s / dt, 4)         # the voxelizer's share of the wall time
            return o, dt

        host, _ = leg("host", frames, chunk)
        torch.cuda.empty_cache()
        try:
            gpu, _ = leg("auto", frames_gpu, chunk_gpu, ramp_gpu)
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
            # the HOST side of one chunk alone: frame headers parsed, record bytes copied from the file (page cache) into pinned memory by the
            # library's threads, the upload -- what a chunk costs whatever the GPU does (the feed's period is the largest of these, the device
            # decode beside the voxelizer and the voxelization itself)
            lo_hi = {}
            lo, hi = xtc.byte_range(fn, sel, N)
            pinned = torch.empty(hi - lo + xtc.XTC_PAD, dtype=torch.uint8, pin_memory=True)
            path = os.fsencode(fn)
            _lib._check(lib.mkamd_xtc_copy_bytes(path, lo, hi, pinned.data_ptr(), 0))
            t0 = time.perf_counter()
            for _ in range(3):
                _lib._check(lib.mkamd_xtc_copy_bytes(path, lo, hi, pinned.data_ptr(), 0))
            t_copy = (time.perf_counter() - t0) / 3
            t0 = time.perf_counter()
            for _ in range(3):                                                  # (as the feed does it: the range from the index, the headers out of the copy)
                xtc.byte_range(fn, sel, N)
                xtc.chunk_desc_mem(fn, sel, N, pinned.data_ptr(), lo, hi)
            lo_hi["header_parse_ms_per_call"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            t0 = time.perf_counter()
            xtc.chunk_desc(fn, sel, N)
            lo_hi["header_parse_from_the_file_ms_per_call"] = round((time.perf_counter() - t0) * 1e3, 3)
            d_up = torch.empty_like(pinned, device=dev)
            d_up.copy_(pinned, non_blocking=True)
            e0.record(st); d_up.copy_(pinned, non_blocking=True); d_up.copy_(pinned, non_blocking=True); e1.record(st)
            torch.cuda.synchronize(dev)
            t_up = e0.elapsed_time(e1) / 2 * 1e-3
            gpu.update(lo_hi, host_byte_copy_ms_per_call=round(t_copy * 1e3, 3), host_byte_copy_GBs=round((hi - lo) / t_copy / 1e9, 1),
                       upload_ms_per_call=round(t_up * 1e3, 3), upload_GBs=round((hi - lo) / t_up / 1e9, 1),
                       voxelizer_ms_per_call=round(chunk_gpu * per_frame_s * 1e3, 3) if per_frame_s else None)
            del pinned, d_up
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
        if (steady or fps) * per_frame_s > 0.9 or "frames_per_s" not in gpu:
            out["bottleneck"] = "GPU (voxelizer)" if (steady or fps) * per_frame_s > 0.9 else "host XTC decode"
        else:
            # which stage of a chunk is the longest (each measured alone above; they overlap in the feed)
            stages = {"host byte copy (file -> pinned memory)": gpu.get("host_byte_copy_ms_per_call", 0.0) + gpu.get("header_parse_ms_per_call", 0.0),
                      "upload (PCIe)": gpu.get("upload_ms_per_call", 0.0),
                      "device XTC decode": gpu.get("decode_kernels_ms_per_call", 0.0),
                      "GPU (voxelizer)": gpu.get("voxelizer_ms_per_call") or 0.0}
            worst = max(stages, key=stages.get)
            period = 1e3 * chunk_gpu / (steady or fps)
            out["bottleneck"] = (f"{worst}: {stages[worst]:.1f} ms of a {period:.1f}-ms period per {chunk_gpu} frames" if worst != "GPU (voxelizer)" else
                                 f"GPU: the voxelizer ({stages[worst]:.1f} ms alone) with the next chunk's decode ({stages['device XTC decode']:.1f} ms alone) and "
                                 f"pre-pass beside it on the same CUs = a {period:.1f}-ms period per {chunk_gpu} frames; host {stages['host byte copy (file -> pinned memory)']:.1f} ms "
                                 f"and upload {stages['upload (PCIe)']:.1f} ms per chunk run beside both")
            out["stage_ms_per_call"] = {k: round(v, 2) for k, v in stages.items()}
    torch.cuda.empty_cache()
    return out
