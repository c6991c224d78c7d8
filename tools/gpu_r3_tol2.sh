mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for tol in 0 1e-6; do
  (timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single --value-tol $tol > gpurun_out/tol2_${tol}_pipe$rep.log 2>&1)
  (timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single --no-pipeline --value-tol $tol > gpurun_out/tol2_${tol}_nopipe$rep.log 2>&1)
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/tol2_*.log')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l)
            print(f.split('/')[-1], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
PY
