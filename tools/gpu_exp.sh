mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in libmkamd.so libmkamd_e512.so libmkamd_e768.so; do
  for wl in cfg2 cfg5; do
  MKAMD_LIB=$GRAFT_REPO_ROOT/moleculekit_amd/csrc/$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > gpurun_out/exp_${lib}_$wl.log 2>&1
  done
done
