# tools/gpu_variants.sh -- same-box comparison of several builds of the library (cfg2, in order and pipelined): $@ = variant .so files
mkdir -p gpurun_out
for rep in 1 2; do
for lib in moleculekit_amd/csrc/libmkamd.so "$@"; do
  tag=$(basename $lib .so)
  (MKAMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python bench.py --no-cpu-baseline --no-extra --no-pipeline > gpurun_out/var_${tag}_nopipe$rep.log 2>&1)
  (MKAMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/var_${tag}_pipe$rep.log 2>&1)
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/var_*.log')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l)
            print(f.split('/')[-1][4:-4].ljust(34), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
PY
