# tools/gpu_batch_sweep.sh -- cfg2 throughput against the number of grids per call (pipelined and in order)
for b in 256 512 1024; do for mode in "" "--no-pipeline"; do
python bench.py --no-cpu-baseline --no-extra --no-single --batch $b $mode 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $b $mode'.ljust(28), 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'], 'sustained', (d.get('sustained') or {}).get('value'))
"
done; done
