mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for wl in cfg1 cfg2 cfg3; do
  for t in -1 0 1 2; do
    timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $wl --lds-tier $t > gpurun_out/bench_${wl}_t$t.log 2>&1
  done
done
for f in gpurun_out/bench_cfg[123]_t*.log; do echo "== $f"; tail -1 $f | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])
except Exception as e: print('ERR', e)
"; done
