mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
for wl in cfg2 cfg3 cfg1 cfg5; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl --no-pipeline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done
done
