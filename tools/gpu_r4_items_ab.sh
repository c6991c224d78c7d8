# same-box A/B of the workgroup-per-item kernel (cfg3 / cfg5): in-tree and .variants/libmkamd_<name>.so, three rounds
export TMPDIR=/tmp
for rep in 1 2 3; do
for n in intree "$@"; do
  lib=$GRAFT_REPO_ROOT/.variants/libmkamd_$n.so; [ $n = intree ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  for wl in cfg3 cfg5; do
  (MKAMD_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-extra --min-seconds 0 --no-single --steps 20 --warmup 4 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n $wl'.ljust(24), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
")
  done
done
done 2>&1 | tee gpurun_out/r4_items_ab.txt
