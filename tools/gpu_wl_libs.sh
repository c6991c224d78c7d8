# tools/gpu_wl_libs.sh "wl..." LIB... -- same-box: several library builds on the given workloads (bench.py secondary lines), two rounds
mkdir -p gpurun_out
export TMPDIR=/tmp
WLS=$1; shift
for rep in 1 2; do
for wl in $WLS; do
for l in "$@"; do
  (MKAMD_LIB=$GRAFT_REPO_ROOT/$l timeout 300 python bench.py --no-cpu-baseline --no-extra --workload $wl 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl', '$l'.ljust(44), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
")
done
done
done
