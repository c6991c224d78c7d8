# tools/gpu_kstats.sh [bench args] -- rocprofv3 kernel stats of the in-order cfg2 step with the in-tree library + a quick parity run
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/ks_B
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_B -- python $R/bench.py --no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2 --no-pipeline "$@" > $R/gpurun_out/ks_B.log 2>&1)
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/ks_B/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.25: print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(4), f'{float(r["AverageNs"]) / 1e3:9.1f} us')
PY
(timeout 300 python bench.py --no-cpu-baseline --no-extra --no-single --min-seconds 0 --no-pipeline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in order: ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'])")
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2)
