mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in cfg2 cfg3 cfg5 cfg4; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > gpurun_out/bench_$wl.log 2>&1
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline > gpurun_out/bench_cfg2_nopipe.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_cfg2.log 2>&1)
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
