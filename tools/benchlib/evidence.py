"""What the bench lines quote from the committed profiles (PMC passes stamped with the build they were taken on) and the roofline objects."""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np

from .workloads import DEFAULT_BATCH, FP32_VALU_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, algorithmic_bytes, algorithmic_flops

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
