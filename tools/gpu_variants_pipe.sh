# tools/gpu_variants_pipe.sh name... -- same-box: the in-tree library and .variants/libmkamd_<name>.so, pipelined and in order, two rounds
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for n in intree "$@"; do
  lib=$GRAFT_REPO_ROOT/.variants/libmkamd_$n.so; [ $n = intree ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  for mode in "" "--no-pipeline"; do
  (MKAMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single $mode 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n $mode'.ljust(28), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'])
")
  done
done
done
