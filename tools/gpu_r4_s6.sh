mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_r4_tile_ab.sh "$@"
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bench_default3.log 2>&1; echo "rc=$?" >> gpurun_out/r4_bench_default3.log)
python - <<'PY'
import json
for l in open('gpurun_out/r4_bench_default3.log'):
    if l.startswith('{'):
        d=json.loads(l); o=d['other_workloads']
        print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'single', d['single_grid_latency_us'], 'dropin', d.get('dropin_call_ms'), d.get('secondary_error'))
        print('stream_cfg4', o.get('stream_cfg4')); print('xtc_cfg4', o.get('xtc_cfg4')); print('cfg4', o['cfg4']['ms_per_step'])
PY
