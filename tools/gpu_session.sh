# round 6, session 7: k_tail with the batched, listed exact_recompute (8 ions: 0.41-0.5 ms at first, 0.30 with the pre-test alone)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_random.py -m gpu -q -x 2>&1 | tail -3)
(timeout 900 python tools/topology_wide_ab.py > gpurun_out/s7_topology_wide_ab.txt 2>&1); cat gpurun_out/s7_topology_wide_ab.txt
for only in 8 300; do
rm -rf gpurun_out/prof_topo_wide_$only
(cd /tmp && AB_ONLY=$only timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_topo_wide_$only -- python $R/tools/topology_wide_ab.py > $R/gpurun_out/s7_rocprof_topo_wide_$only.log 2>&1)
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof_topo_wide_$only/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_tail" in r["Name"]:
            print("$only wide atoms:", r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
find gpurun_out -name "*_kernel_trace.csv" -size +1M -delete
