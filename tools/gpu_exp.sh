mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in cfg1 cfg2 cfg4; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/x_${wl}.log 2>&1
  MKAMD_LIB=$R/.variants/libmkamd_nv52.so timeout 300 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/x_${wl}_nv52.log 2>&1
done
for f in gpurun_out/x_cfg[124].log gpurun_out/x_cfg*_nv52.log; do echo "== $f"; tail -1 $f | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])
except Exception as e: print('ERR', e)
"; done
