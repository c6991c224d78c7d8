# tools/gpu_kstats_ab.sh variant.so [bench args] -- rocprofv3 kernel stats of the in-order cfg2 step for a variant library (A) and the in-tree one (B)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A=$R/$1; shift
for v in A B; do
  lib=$A; [ $v = B ] && lib=$R/moleculekit_amd/csrc/libmkamd.so
  rm -rf gpurun_out/ks_$v
  (cd /tmp && MKAMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$v -- python $R/bench.py --no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2 --no-pipeline "$@" > $R/gpurun_out/ks_$v.log 2>&1)
  echo "== $v"
  python - $v <<'PY'
import csv, glob, sys
f = sorted(glob.glob(f"gpurun_out/ks_{sys.argv[1]}/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.25: print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(4), f'{float(r["AverageNs"]) / 1e3:9.1f} us')
PY
done
