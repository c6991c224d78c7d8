# tools/gpu_round.sh -- one GPU-box session: smoke, parity tests, benches, rocprof (outputs -> gpurun_out/)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log)
(timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
for wl in cfg2 cfg3 cfg5 cfg4; do
  (timeout 400 python bench.py --steps 10 --warmup 2 --workload $wl $([ $wl = cfg2 ] || echo --no-cpu-baseline) > gpurun_out/bench_$wl.log 2>&1; echo "rc=$?" >> gpurun_out/bench_$wl.log)
done
(MKAMD_FORCE_GENERAL=1 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg2_general.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_general.log)
(timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --tile-k 4 > gpurun_out/bench_cfg2_k4.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_k4.log)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_cfg2.log 2>&1; echo "rc=$?" >> $R/gpurun_out/rocprof_cfg2.log)
tail -3 gpurun_out/smoke.log; tail -8 gpurun_out/pytest_gpu.log
for f in gpurun_out/bench_*.log; do echo "== $f"; tail -2 $f | cut -c1-400; done
find gpurun_out/prof_cfg2 -name "*stats*" | head
