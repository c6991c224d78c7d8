# round 6, session 61: k_exact_redo (the exact cut-off hits of a topology call recomputed by many waves): GPU tests, A-B, k_tail / k_exact_redo durations
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s61_tests.txt
timeout 900 python tools/topology_wide_ab.py 2>&1 | grep -v amdgpu | tee gpurun_out/topology_wide_ab.txt
for nw in 8 300; do
  (cd /tmp && AB_ONLY=$nw timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wide_$nw -- python $GRAFT_REPO_ROOT/tools/topology_wide_ab.py > /dev/null 2>&1)
  python - <<PY
import csv, glob
import os; f = sorted(glob.glob("gpurun_out/prof_wide_$nw/*/*_kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "k_tail" in r["Name"] or "k_exact" in r["Name"] or "k_zero" in r["Name"]:
        print("$nw ions:", r["Name"][:60], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1))
PY
done 2>&1 | tee gpurun_out/topology_wide_kernels.txt
