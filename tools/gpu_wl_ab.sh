# tools/gpu_wl_ab.sh base.so wl... -- same-box A/B of the in-tree build against another one on the given bench workloads
A=$GRAFT_REPO_ROOT/$1; shift
mkdir -p gpurun_out
for rep in 1 2; do for wl in "$@"; do for v in base new; do
  lib=$A; [ $v = new ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  MKAMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra --workload $wl > gpurun_out/w_${wl}_${v}_$rep.log 2>&1
done; done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/w_*.log')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(f.split('/')[-1], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'], 'value', d['value'])
PY
