mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -2
for so in 0 1; do
for wl in cfg4 cfg2; do
MKAMD_SPATIAL_ORDER=$so timeout 300 python bench.py --no-cpu-baseline --no-pipeline --workload $wl 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order=$so $wl', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done
done
