# tools/gpu_wl_ab2.sh variant.so wl... -- same-box A/B of a variant library (A) against the in-tree one (B) on given workloads
mkdir -p gpurun_out
export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/$1; shift
for rep in 1 2 3; do
for wl in "$@"; do
for v in A B; do
  lib=$A; [ $v = B ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  (MKAMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra --workload $wl > gpurun_out/ab2_${wl}_${v}$rep.log 2>&1)
done
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab2_*.log')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l)
            print(f.split('/')[-1], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
PY
