mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for wl in cfg3 cfg5; do
(timeout 400 python bench.py --workload $wl > gpurun_out/bench_$wl.log 2>&1; echo "rc=$?" >> gpurun_out/bench_$wl.log)
tail -2 gpurun_out/bench_$wl.log | cut -c1-400
done
cd /tmp
for wl in cfg3 cfg5; do
rm -rf $R/gpurun_out/prof_$wl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$wl -- python $R/bench.py --no-cpu-baseline --workload $wl > $R/gpurun_out/rocprof_$wl.log 2>&1
done
