mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for wl in cfg1 cfg3 cfg4 cfg5; do
rm -rf $R/gpurun_out/prof_$wl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$wl -- python $R/bench.py --no-cpu-baseline --workload $wl > $R/gpurun_out/rocprof_$wl.log 2>&1
f=$(ls $R/gpurun_out/prof_$wl/*/*_kernel_stats.csv | head -1)
echo "== $wl"; head -6 $f | cut -c1-60,200-
done
