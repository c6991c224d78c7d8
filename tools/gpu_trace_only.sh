mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
TRACE="--no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 40 --warmup 5"
rm -rf gpurun_out/prof_cfg2 gpurun_out/prof_cfg2_nopipe
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2 -- python $R/bench.py $TRACE --workload cfg2 > $R/gpurun_out/rocprof_cfg2.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2_nopipe -- python $R/bench.py $TRACE --no-pipeline > $R/gpurun_out/rocprof_cfg2_nopipe.log 2>&1)
grep -h '"metric"' gpurun_out/rocprof_cfg2.log gpurun_out/rocprof_cfg2_nopipe.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
