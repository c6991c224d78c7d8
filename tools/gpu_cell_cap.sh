# tools/gpu_cell_cap.sh -- in-order cfg2 step against the record slots per cell of the direct layout (MKAMD_CELL_CAP, A-B knob)
for rep in 1 2; do for cap in 128 96 80 64; do
MKAMD_CELL_CAP=$cap python bench.py --no-cpu-baseline --no-extra --no-single --min-seconds 0 --no-pipeline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('cap $cap'.ljust(10), 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'])
"
done; done
