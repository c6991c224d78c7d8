"""Copy the judged summaries of tools/gpu_evidence.sh from gpurun_out/ (scratch) into profiles/<tag>_* (tracked; tag = argv[1], default r5) and stamp every
PMC summary with the source hash of the library it was taken on (moleculekit_amd._build.built_hash(): bench.py refuses
counters of another build).  Runs on the GPU box at the end of the evidence session, and again here (idempotent)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from moleculekit_amd import _build  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
SRC = _build.built_hash()
os.makedirs("profiles", exist_ok=True)
PROF = "--no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2"
for name in ("cfg2", "cfg2_nopipe", "torchrun1", "dist", "dist_torchrun1"):
    f = f"gpurun_out/bench_{name}.log"
    if os.path.exists(f):
        for line in open(f):
            if line.startswith("{"):
                d = json.loads(line)
                d["_library_src"] = SRC
                err = f"gpurun_out/bench_{name}.err"
                if os.path.exists(err):                      # the torchrun passes run with NCCL_DEBUG=WARN: what RCCL had to say
                    warn = [l.strip()[:300] for l in open(err, errors="replace") if "NCCL WARN" in l or "NCCL ERROR" in l]
                    d["_nccl_debug_warn_lines"] = warn[:20]
                    d["_nccl_debug_warn_count"] = len(warn)
                json.dump(d, open(f"profiles/{tag}_bench_{name}.json", "w"), indent=1)
                break
# the distance lines of the TRACED passes (rocprofv3 --kernel-trace --stats with the burns): the fractions and clocks the stats files are compared with
for log, what in (("rocprof_dist.log", "dist"), ("rocprof_reduction.log", "reduction")):
    f = "gpurun_out/" + log
    if os.path.exists(f):
        for line in open(f, errors="replace"):
            if line.startswith("{"):
                d = json.loads(line)
                d["_library_src"] = SRC
                d["_note"] = "the bench line printed by the pass that profiles/%s_%s_rocprofv3_kernel_stats.csv was taken on" % (tag, what)
                json.dump(d, open(f"profiles/{tag}_bench_{what}_traced_pass.json", "w"), indent=1)
                break
for d in ("cfg2", "cfg2_nopipe", "cfg1", "cfg3", "cfg4", "cfg4_plain", "cfg5", "dist", "reduction"):
    stats = sorted(glob.glob(f"gpurun_out/prof_{d}/*/*_kernel_stats.csv"), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], f"profiles/{tag}_{d}_rocprofv3_kernel_stats.csv")


def collect(passes, out_name, items, cmd, note_extra=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in passes:
        fs = sorted(glob.glob(f"gpurun_out/pmc_{d}/*/*counter_collection.csv"), key=os.path.getmtime)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[-1])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        return None
    out = {"_library_src": SRC, "_items_per_launch": items, "_full_batch_launches_only": True,
           "_command": f"rocprofv3 --kernel-trace --pmc <counters of one pass> -- python bench.py {cmd}".strip(),
           "_note": "mean per launch; one rocprofv3 --pmc pass per counter group (tools/gpu_evidence.sh), FETCH_SIZE and WRITE_SIZE in "
                    "passes of their own; KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950: double it (MI355X_MICROARCH.md, HBM). "
                    "--no-single: no single-grid probe launches, every launch of a kernel is one full step. GRBM_GUI_ACTIVE: shader "
                    "cycles of the launch summed over the 8 XCDs (counter passes run every kernel ALONE: divide by a duration of the same pass, not of a trace run in which kernels overlap)." + note_extra}
    for k in sorted(acc):
        out[k] = {c: round(sum(v) / len(v), 2) for c, v in sorted(acc[k].items())}
        out[k]["_launches"] = max(len(v) for v in acc[k].values())
    json.dump(out, open(f"profiles/{out_name}", "w"), indent=1)
    return out


def report(out_name, out, alg_mb):
    hot = [k for k in out if isinstance(out[k], dict) and ("k_voxelize_tiles" in k or "k_voxelize_items" in k or "k_dist_pairs" in k or "k_dist_rows" in k or "k_dist_frame" in k)]
    n = max((out[k]["_launches"] for k in hot), default=0)
    for k in hot:
        v = out[k]
        if "SQ_INSTS_VALU" in v and v.get("SQ_WAVES"):
            clk = v["GRBM_GUI_ACTIVE"] / 8.0 if v.get("GRBM_GUI_ACTIVE") else v.get("SQ_BUSY_CYCLES", 0) / 32.0     # (summed over 8 XCDs / 32 SEs)
            busy = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / clk if clk else float("nan")
            print(f"{out_name}: {k[:52]:52s} VALU/wave {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:8.1f}  SALU/wave {v.get('SQ_INSTS_SALU', 0) / v['SQ_WAVES']:7.1f}  "
                  f"LDS/wave {v.get('SQ_INSTS_LDS', 0) / v['SQ_WAVES']:6.1f}  VALU-busy {busy:.3f}  launches {v['_launches']}")
    step = 0.0
    for k, v in out.items():
        if isinstance(v, dict) and k.startswith("mkamd::") and "FETCH_SIZE" in v and "WRITE_SIZE" in v and n:
            b = (v["WRITE_SIZE"] + 2 * v["FETCH_SIZE"]) * 1024 * v["_launches"] / n
            step += b
            print(f"   {k[:52]:52s} {b / 1e6:9.1f} MB per step  (read {2 * v['FETCH_SIZE'] * 1024 * v['_launches'] / n / 1e6:8.1f}, written {v['WRITE_SIZE'] * 1024 * v['_launches'] / n / 1e6:8.1f})")
    if step:
        print(f"   step total {step / 1e6:.1f} MB (algorithmic {alg_mb:.1f} MB)")


ALG = {"cfg2": 2710.7, "cfg1": 2107.3, "cfg3": 14582.0, "cfg4": 1243.9, "cfg5": 29092.0, "dist": 836.4}
for wl in ("cfg2", "cfg1", "cfg3", "cfg4", "cfg5"):
    passes = [f"{wl}_sq1", f"{wl}_fetch", f"{wl}_write"] + ([f"{wl}_sq2"] if wl == "cfg2" else [])
    o = collect(passes, f"{tag}_{wl}_pmc_counters.json", bench.DEFAULT_BATCH[wl], f"{PROF} --workload {wl}")
    if o:
        report(f"{tag}_{wl}", o, ALG[wl])
o = collect(("cfg2_nopipe_sq1", "cfg2_nopipe_fetch", "cfg2_nopipe_write"), f"{tag}_cfg2_nopipe_pmc_counters.json", bench.DEFAULT_BATCH["cfg2"], f"{PROF} --no-pipeline")
if o:
    report(f"{tag}_cfg2_nopipe", o, ALG["cfg2"])
for mode in ("periodic", "nonperiodic"):
    o = collect([f"dist_{mode}_{p}" for p in ("sq1", "fetch", "write")],
                f"{tag}_dist_pmc_counters.json" if mode == "periodic" else f"{tag}_dist_nonperiodic_pmc_counters.json",
                bench.DEFAULT_BATCH["dist"], f"--workload dist --no-cpu-baseline --steps 8 --warmup 2 (MKAMD_DIST_ONLY={mode})")
    if o:
        report(f"{tag}_dist_{mode}", o, ALG["dist"])
o = collect(("reduction_sq1", "reduction_sq2"), f"{tag}_reduction_pmc_counters.json", 512,
            "--workload dist --no-cpu-baseline --settle-seconds 0 --steps 4 --warmup 1 (MKAMD_DIST_ONLY=reduction)",
            note_extra="  The group-reduction leg: 200 groups x 15 atoms x 512 frames, all 19 900 group pairs, periodic (tools/benchlib/distances.py::bench_reductions).")
if o:
    for k, v in o.items():
        if isinstance(v, dict) and "k_dist_reduction_closest" in k and "SQ_INSTS_VALU" in v:
            pairs = 200 * 199 // 2 * 225 * 512
            clk = v["GRBM_GUI_ACTIVE"] / 8.0
            print(f"{tag}_reduction: {k[:60]} VALU wave-instructions {v['SQ_INSTS_VALU']:.0f} = {v['SQ_INSTS_VALU'] * 64 / pairs:.2f} lane-instructions per atom pair; "
                  f"VALU-busy {v['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / clk:.3f}")
for src, dst in (("single_latency.txt", "single_latency.txt"), ("dropin_profile.txt", "dropin_profile.txt"), ("graph_latency.txt", "graph_latency.txt"),
                 ("reduction_probe.txt", "reduction_probe.txt"), ("reduction_few_probe.txt", "reduction_few_probe.txt"), ("topology_wide_ab_final.txt", "topology_wide_ab_final.txt"),
                 ("xtc_gpu_probe.txt", "xtc_gpu_probe.txt"), ("dist_shapes_probe.txt", "dist_shapes_probe.txt"), ("random_sweeps.txt", "random_sweeps_final.txt"),
                 ("sqrt_exact.txt", "sqrt_exact.txt")):
    f = f"gpurun_out/{src}"
    if os.path.exists(f):
        with open(f) as fh:
            text = "".join(l for l in fh if "amdgpu.ids" not in l)
        open(f"profiles/{tag}_{dst}", "w").write(text)
print("\n".join(sorted(f for f in os.listdir("profiles") if f.startswith(tag))))
