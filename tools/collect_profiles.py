"""Copy the judged summaries of a GPU session from gpurun_out/ (scratch) into profiles/ (tracked).

usage: python tools/collect_profiles.py [round_tag]      (default r1)
  bench_*.log                 -> profiles/<tag>_bench_<name>.json   (the JSON line only)
  prof_cfg2*/.../kernel_stats -> profiles/<tag>_cfg2[_nopipe]_rocprofv3_kernel_stats.csv
  timeline_prof_*.txt         -> profiles/<tag>_cfg2[_nopipe]_timeline.txt
  pmc_*/counter_collection    -> profiles/<tag>_cfg2_pmc_counters.json (per kernel, mean per launch)
  ubench_mem.txt, bench_distance.log
"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
os.makedirs("profiles", exist_ok=True)
for f in sorted(glob.glob("gpurun_out/bench_*.log")):
    name = os.path.basename(f)[len("bench_"):-len(".log")]
    for line in open(f):
        if line.startswith("{"):
            json.dump(json.loads(line), open(f"profiles/{tag}_bench_{name}.json", "w"), indent=1)
            break
for d, suffix in (("prof_cfg2", ""), ("prof_cfg2_nopipe", "_nopipe")):
    stats = sorted(glob.glob(f"gpurun_out/{d}/*/*_kernel_stats.csv"), key=os.path.getmtime)    # gpurun merges: keep the newest
    if stats:
        shutil.copy(stats[-1], f"profiles/{tag}_cfg2{suffix}_rocprofv3_kernel_stats.csv")
    t = f"gpurun_out/timeline_{d}.txt"
    if os.path.exists(t):
        shutil.copy(t, f"profiles/{tag}_cfg2{suffix}_timeline.txt")
for wl in ("cfg1", "cfg3", "cfg4", "cfg5"):                 # rocprofv3 --kernel-trace --stats of the other workloads, when taken
    stats = sorted(glob.glob(f"gpurun_out/prof_{wl}/*/*_kernel_stats.csv"), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], f"profiles/{tag}_{wl}_rocprofv3_kernel_stats.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("gpurun_out/pmc_*/")):
    # gpurun merges every call's output into the same directories: keep the newest pass only
    fs = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc:
    sys.path.insert(0, os.getcwd())
    import bench
    out = {"_items_per_launch": bench.DEFAULT_BATCH["cfg2"],
           "_note": "mean per launch over the launches of `python bench.py --steps 4 --warmup 1 --no-pipeline` "
                    "(one rocprofv3 --pmc pass per counter group, tools/gpu_pmc.sh); FETCH_SIZE / WRITE_SIZE in KiB "
                    "(FETCH_SIZE counts 64 B per 128-B request on gfx950: double it, MI355X_MICROARCH.md HBM section)"}
    for k in sorted(acc):
        out[k] = {c: round(sum(v) / len(v), 2) for c, v in sorted(acc[k].items())}
        out[k]["_launches"] = max(len(v) for v in acc[k].values())
    json.dump(out, open(f"profiles/{tag}_cfg2_pmc_counters.json", "w"), indent=1)
for src, dst in (("ubench_mem.txt", f"{tag}_ubench_mem.txt"),):
    if os.path.exists("gpurun_out/" + src):
        shutil.copy("gpurun_out/" + src, "profiles/" + dst)
for src, dst in (("bench_xtc.txt", f"{tag}_bench_xtc.txt"), ("host_paths.txt", f"{tag}_host_paths.txt"), ("pmc_dist.txt", f"{tag}_dist_pairs_pmc.txt"),
                 ("kstats_reduction.txt", f"{tag}_dist_reduction_kernel_stats.txt"), ("diag_breakdown.txt", f"{tag}_cfg2_valu_breakdown.txt"),
                 ("latency_probe.txt", f"{tag}_latency_probe.txt"), ("phase_timers.txt", f"{tag}_phase_timers.txt"),
                 ("dropin_profile.txt", f"{tag}_dropin_host_profile.txt"), ("pmc_bin.txt", f"{tag}_prepass_instruction_counts.txt")):
    if os.path.exists("gpurun_out/" + src):
        shutil.copy("gpurun_out/" + src, "profiles/" + dst)
for wl in ("cfg1", "cfg3", "cfg4", "cfg5", "dist"):
    if os.path.exists(f"gpurun_out/{wl}_pmc_counters.json"):
        shutil.copy(f"gpurun_out/{wl}_pmc_counters.json", f"profiles/{tag}_{wl}_pmc_counters.json")
if os.path.exists("gpurun_out/single_timeline_cfg2.txt") and os.path.exists("gpurun_out/single_timeline_3ptb.txt"):
    open(f"profiles/{tag}_single_call_timeline.txt", "w").write(open("gpurun_out/single_timeline_cfg2.txt").read() + open("gpurun_out/single_timeline_3ptb.txt").read())
print("\n".join(sorted(os.listdir("profiles"))))
