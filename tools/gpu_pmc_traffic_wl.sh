# tools/gpu_pmc_traffic_wl.sh wl... -- HBM traffic of the tile / item kernel on other workloads: FETCH_SIZE and WRITE_SIZE in their
# own rocprofv3 --pmc passes (the MI355X guide's recipe), written to gpurun_out/<wl>_pmc_counters.json in the layout of
# profiles/r*_cfg2_pmc_counters.json (bench.py's pmc_traffic reads profiles/r*_<wl>_pmc_counters.json)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcw_${wl}_$c
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcw_${wl}_$c -- python $R/bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/pmcw_${wl}_$c.log 2>&1)
  done
done
cd $R
python - "$@" <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
for wl in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = sorted(glob.glob(f"gpurun_out/pmcw_{wl}_{c}/*/*counter_collection.csv"), key=os.path.getmtime)
        for r in csv.DictReader(open(fs[-1])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"_items_per_launch": bench.DEFAULT_BATCH[wl],
           "_note": f"mean per launch of `python bench.py --workload {wl} --steps 4 --warmup 1 --no-pipeline` (one rocprofv3 --pmc pass per "
                    "counter, tools/gpu_pmc_traffic_wl.sh); KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950: double it"}
    for k in sorted(acc):
        big = {c: [x for x in v if x >= 0.5 * max(v)] for c, v in acc[k].items()}      # the full-batch launches (not the single-grid probes)
        out[k] = {c: round(sum(v) / len(v), 2) for c, v in sorted(big.items())}
        out[k]["_launches"] = max(len(v) for v in big.values())
    json.dump(out, open(f"gpurun_out/{wl}_pmc_counters.json", "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict) and ("voxelize" in k):
            print(wl, k[-40:], v)
PY
