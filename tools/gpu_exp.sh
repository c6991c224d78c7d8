mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for wl in cfg1 cfg2 cfg3 cfg4 cfg5; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $wl > gpurun_out/bench_${wl}_new.log 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $wl --no-pipeline > gpurun_out/bench_${wl}_new_nopipe.log 2>&1
done
for f in gpurun_out/bench_cfg*_new*.log; do echo "== $f"; tail -1 $f | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])
except Exception as e: print('ERR', e)
"; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_cfg2np -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-pipeline > $R/gpurun_out/trace_cfg2np.log 2>&1)
f=$(find gpurun_out/trace_cfg2np -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 10 1
