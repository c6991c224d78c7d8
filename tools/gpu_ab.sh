# tools/gpu_ab.sh -- same-box A/B of two builds of the library: $1 = variant .so (A), the in-tree library is B
mkdir -p gpurun_out
export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/$1
for rep in 1 2; do
for v in A B; do
  lib=$A; [ $v = B ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  (MKAMD_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-pipeline ${AB_ARGS:-} > gpurun_out/ab_${v}_nopipe$rep.log 2>&1; echo "rc=$?" >> gpurun_out/ab_${v}_nopipe$rep.log)
  (MKAMD_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab_${v}_pipe$rep.log 2>&1; echo "rc=$?" >> gpurun_out/ab_${v}_pipe$rep.log)
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_*.log')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l)
            print(f.split('/')[-1], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'], 'single', d.get('single_grid_latency_us'))
            for k,v in d.get('other_workloads',{}).items():
                print('     ',k, v['ms_per_step'], v['roofline']['kernel_avg_ms'], v['roofline']['frac'])
PY
