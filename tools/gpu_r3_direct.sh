# tools/gpu_r3_direct.sh -- direct binning (k_bin_direct) against the count / scan / fill chain: parity tests, then same-box
# timings in order and pipelined, then the in-order kernel stats of both
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
for rep in 1 2; do
for d in 0 1; do
  (MKAMD_DIRECT=$d timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single > gpurun_out/dir_${d}_pipe$rep.log 2>&1)
  (MKAMD_DIRECT=$d timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single --no-pipeline > gpurun_out/dir_${d}_nopipe$rep.log 2>&1)
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/dir_*.log')):
    ok = False
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); ok = True
            print(f.split('/')[-1], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
    if not ok: print(f, 'FAILED', open(f).read()[-500:])
PY
for d in 0 1; do
  rm -rf gpurun_out/ks_d$d
  (cd /tmp && MKAMD_DIRECT=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_d$d -- python $R/bench.py --no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2 --no-pipeline > $R/gpurun_out/ks_d$d.log 2>&1)
  echo "== in order, MKAMD_DIRECT=$d"
  python - $d <<'PY'
import csv, glob, sys
f = sorted(glob.glob(f"gpurun_out/ks_d{sys.argv[1]}/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.05: print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(4), f'{float(r["AverageNs"]) / 1e3:9.1f} us')
PY
done
