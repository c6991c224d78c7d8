mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for wl in cfg1 cfg2 cfg3 cfg4 cfg5; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/x_${wl}.log 2>&1
  timeout 300 python bench.py --no-cpu-baseline --workload $wl --no-pipeline > gpurun_out/x_${wl}_nopipe.log 2>&1
done
for f in gpurun_out/x_cfg?.log gpurun_out/x_cfg?_nopipe.log; do echo "== $f"; tail -1 $f | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])
except Exception as e: print('ERR', e)
"; done
