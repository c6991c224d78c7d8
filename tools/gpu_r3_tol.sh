# tools/gpu_r3_tol.sh -- same-box: exact mode vs tolerance-aware reach (eps 1e-6, 5e-6), default build and the build
# without packed f32 (.variants/libmkamd_nopk.so); GPU parity tests on both builds first
mkdir -p gpurun_out
export TMPDIR=/tmp
NOPK=$GRAFT_REPO_ROOT/.variants/libmkamd_nopk.so
(timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
(MKAMD_LIB=$NOPK timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_nopk.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_nopk.log)
tail -3 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu_nopk.log
for rep in 1 2; do
for tol in 0 1e-6 5e-6; do
  (timeout 300 python bench.py --no-cpu-baseline --no-pipeline --value-tol $tol > gpurun_out/tol_${tol}_nopipe$rep.log 2>&1)
  (timeout 300 python bench.py --no-cpu-baseline --no-extra --value-tol $tol > gpurun_out/tol_${tol}_pipe$rep.log 2>&1)
  (MKAMD_LIB=$NOPK timeout 300 python bench.py --no-cpu-baseline --no-extra --no-pipeline --value-tol $tol > gpurun_out/tolnopk_${tol}_nopipe$rep.log 2>&1)
  (MKAMD_LIB=$NOPK timeout 300 python bench.py --no-cpu-baseline --no-extra --value-tol $tol > gpurun_out/tolnopk_${tol}_pipe$rep.log 2>&1)
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/tol*.log')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l)
            print(f.split('/')[-1], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
            for k,v in d.get('other_workloads',{}).items():
                if 'roofline' in v: print('     ',k, v['ms_per_step'], v['roofline']['kernel_avg_ms'], v['roofline']['frac'])
PY
