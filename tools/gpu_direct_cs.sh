# tools/gpu_direct_cs.sh LIB... -- in-order kernel stats with direct binning on (MKAMD_DIRECT=1) for several builds
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in "$@"; do
  n=$(basename $l .so)
  rm -rf gpurun_out/kd_$n
  (cd /tmp && MKAMD_LIB=$R/$l MKAMD_DIRECT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kd_$n -- python $R/bench.py --no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2 --no-pipeline > $R/gpurun_out/kd_$n.log 2>&1)
  echo "== $n (in order, direct binning on)"; grep '^{' gpurun_out/kd_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   ms/step', d['ms_per_step'])"
  python - $n <<'PY'
import csv, glob, sys
f = sorted(glob.glob(f"gpurun_out/kd_{sys.argv[1]}/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.2: print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(4), f'{float(r["AverageNs"]) / 1e3:9.1f} us')
PY
done
