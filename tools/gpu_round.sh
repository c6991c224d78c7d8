# tools/gpu_round.sh -- one GPU-box session: smoke, parity tests, benches, rocprof (outputs -> gpurun_out/)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log)
(timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
(timeout 400 python bench.py > gpurun_out/bench_cfg2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2.log)
for wl in cfg1 cfg3 cfg4 cfg5; do  # (defaults: see DEFAULT_BATCH in bench.py)
  (timeout 400 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/bench_$wl.log 2>&1; echo "rc=$?" >> gpurun_out/bench_$wl.log)
done
(timeout 400 python bench.py --no-cpu-baseline --no-pipeline > gpurun_out/bench_cfg2_nopipe.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_nopipe.log)
(MKAMD_FORCE_GENERAL=1 timeout 400 python bench.py --no-cpu-baseline --no-pipeline > gpurun_out/bench_cfg2_general.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_general.log)
(timeout 400 python bench.py --no-cpu-baseline --no-pipeline --tile-k 4 > gpurun_out/bench_cfg2_k4.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_k4.log)
(timeout 400 python bench.py --no-cpu-baseline --no-pipeline --workload cfg1 --lds-tier 0 > gpurun_out/bench_cfg1_tier0.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg1_tier0.log)
(timeout 400 python bench.py --no-cpu-baseline --batch 32 > gpurun_out/bench_cfg2_b32.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_b32.log)
rm -rf gpurun_out/prof_cfg2 gpurun_out/prof_cfg2_nopipe
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/rocprof_cfg2.log 2>&1; echo "rc=$?" >> $R/gpurun_out/rocprof_cfg2.log)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2_nopipe -- python $R/bench.py --no-cpu-baseline --no-pipeline > $R/gpurun_out/rocprof_cfg2_nopipe.log 2>&1; echo "rc=$?" >> $R/gpurun_out/rocprof_cfg2_nopipe.log)
for v in prof_cfg2 prof_cfg2_nopipe; do
  f=$(find gpurun_out/$v -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $f 12 2 > gpurun_out/timeline_$v.txt 2>&1
done
(timeout 300 python bench.py --workload dist > gpurun_out/bench_distance.log 2>&1; echo "rc=$?" >> gpurun_out/bench_distance.log)
(timeout 300 python bench.py --workload dropin --steps 50 > gpurun_out/bench_dropin.log 2>&1; echo "rc=$?" >> gpurun_out/bench_dropin.log)
(./.variants/ubench_mem > gpurun_out/ubench_mem.txt 2>&1)
tail -3 gpurun_out/smoke.log; tail -4 gpurun_out/pytest_gpu.log
python tools/summarize.py
