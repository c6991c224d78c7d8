"""Copy the judged summaries of tools/gpu_r3_evidence.sh from gpurun_out/ (scratch) into profiles/r3_* (tracked)."""
import collections, csv, glob, json, os, shutil, sys

tag = "r3"
os.makedirs("profiles", exist_ok=True)
PROF = "--no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2"
for name in ("cfg2", "cfg2_nopipe", "torchrun1"):
    f = f"gpurun_out/bench_{name}.log"
    if os.path.exists(f):
        for line in open(f):
            if line.startswith("{"):
                json.dump(json.loads(line), open(f"profiles/{tag}_bench_{name}.json", "w"), indent=1)
                break
for d in ("cfg2", "cfg2_nopipe", "cfg1", "cfg3", "cfg4", "cfg5"):
    stats = sorted(glob.glob(f"gpurun_out/prof_{d}/*/*_kernel_stats.csv"), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], f"profiles/{tag}_{d}_rocprofv3_kernel_stats.csv")


def collect(passes, out_name, cmd_extra, note_extra=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in passes:
        fs = sorted(glob.glob(f"gpurun_out/pmc_{d}/*/*counter_collection.csv"), key=os.path.getmtime)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[-1])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        return
    sys.path.insert(0, os.getcwd())
    import bench
    out = {"_items_per_launch": bench.DEFAULT_BATCH["cfg2"], "_full_batch_launches_only": True,
           "_command": f"rocprofv3 --kernel-trace --pmc <counters of one pass> -- python bench.py {PROF} {cmd_extra}".strip(),
           "_note": "mean per launch; one rocprofv3 --pmc pass per counter group (tools/gpu_r3_evidence.sh), FETCH_SIZE and WRITE_SIZE in "
                    "passes of their own; KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950: double it (MI355X_MICROARCH.md, HBM). "
                    "--no-single: no single-grid probe launches, every launch of a kernel is one full 256-grid step." + note_extra}
    for k in sorted(acc):
        out[k] = {c: round(sum(v) / len(v), 2) for c, v in sorted(acc[k].items())}
        out[k]["_launches"] = max(len(v) for v in acc[k].values())
    json.dump(out, open(f"profiles/{out_name}", "w"), indent=1)
    tile = [k for k in out if "k_voxelize_tiles" in k and isinstance(out[k], dict)]
    for k in tile:
        v = out[k]
        if "SQ_INSTS_VALU" in v:
            tiles = 131072
            print(out_name, k, "VALU/tile", round(v["SQ_INSTS_VALU"] / tiles), "SALU/tile", round(v.get("SQ_INSTS_SALU", 0) / tiles),
                  "LDS/tile", round(v.get("SQ_INSTS_LDS", 0) / tiles), "launches", v["_launches"])
    step = 0.0
    n = max((out[k]["_launches"] for k in tile), default=0)
    for k, v in out.items():
        if isinstance(v, dict) and k.startswith("mkamd::") and "FETCH_SIZE" in v and "WRITE_SIZE" in v and n:
            b = (v["WRITE_SIZE"] + 2 * v["FETCH_SIZE"]) * 1024 * v["_launches"] / n
            step += b
            print(f"   {k[:48]:48s} {b / 1e6:9.1f} MB per step  (read {2 * v['FETCH_SIZE'] * 1024 * v['_launches'] / n / 1e6:8.1f}, written {v['WRITE_SIZE'] * 1024 * v['_launches'] / n / 1e6:8.1f})")
    if step:
        print(f"   step total {step / 1e6:.1f} MB (algorithmic 2710.7 MB)")


collect(("sq1", "sq2", "fetch", "write"), f"{tag}_cfg2_pmc_counters.json", "")
collect(("nopipe_fetch", "nopipe_write"), f"{tag}_cfg2_nopipe_pmc_traffic.json", "--no-pipeline")
collect(("tol_sq1",), f"{tag}_cfg2_tolerance_pmc_counters.json", "--value-tol 1e-6")
print("\n".join(sorted(f for f in os.listdir("profiles") if f.startswith(tag))))

# the one-molecule call (round 3)
for src, dst in (("single_call_timeline.txt", "single_call_timeline.txt"), ("single_latency.txt", "single_latency.txt"),
                 ("team_probe.txt", "team_probe.txt"), ("gridsync.txt", "gridsync_ubench.txt"), ("dropin_profile.txt", "dropin_profile.txt")):
    f = f"gpurun_out/{src}"
    if os.path.exists(f):
        with open(f) as fh:
            text = "".join(l for l in fh if "amdgpu.ids" not in l)
        open(f"profiles/{tag}_{dst}", "w").write(text)
