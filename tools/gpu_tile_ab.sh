# tools/gpu_r4_tile_ab.sh name... -- round 4, same box: GPU parity tests on the in-tree library, then the in-tree library and
# .variants/libmkamd_<name>.so on the cfg2 step, pipelined and in order, three rounds (tile-kernel ms from the HIP events of the run)
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_pytest_gpu.log)
tail -3 gpurun_out/r4_pytest_gpu.log
for rep in 1 2 3; do
for n in intree "$@"; do
  lib=$GRAFT_REPO_ROOT/.variants/libmkamd_$n.so; [ $n = intree ] && lib=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/libmkamd.so
  for mode in "" "--no-pipeline"; do
  (MKAMD_DIAG=1 MKAMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra --min-seconds 0 --no-single --steps 30 --warmup 5 $mode 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n $mode'.ljust(28), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
")
  done
done
done 2>&1 | tee gpurun_out/r4_tile_ab.txt
