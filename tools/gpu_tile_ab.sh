# tools/gpu_tile_ab.sh name... -- same box, same session: the in-tree library and .variants/libmkamd_<name>.so (tools/build_variant.sh)
# on the cfg2 step, pipelined and in order, three rounds, then once on cfg1 / cfg4 (tile-kernel ms from the library's HIP events)
mkdir -p gpurun_out
export TMPDIR=/tmp MKAMD_ALLOW_DIAGNOSTICS=1
one() {  # label lib args...
  label=$1; lib=$2; shift; shift
  (MKAMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single --steps 30 --warmup 5 "$@" 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$label'.ljust(34), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
")
}
for rep in 1 2 3; do
for n in intree "$@"; do
  lib=$GRAFT_REPO_ROOT/.variants/libmkamd_$n.so; [ $n = intree ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  one "$n" $lib
  one "$n --no-pipeline" $lib --no-pipeline
done
done 2>&1 | tee gpurun_out/tile_ab.txt
for wl in cfg1 cfg4; do
for n in intree "$@"; do
  lib=$GRAFT_REPO_ROOT/.variants/libmkamd_$n.so; [ $n = intree ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  one "$n $wl" $lib --workload $wl
done
done 2>&1 | tee -a gpurun_out/tile_ab.txt
