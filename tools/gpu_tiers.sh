for wl in cfg1 cfg4; do for t in -1 0 1 2; do
python bench.py --no-cpu-baseline --no-extra --no-single --workload $wl --lds-tier $t 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl tier $t', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])
"
done; done
