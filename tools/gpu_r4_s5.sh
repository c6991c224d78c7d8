mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_pytest_gpu.log); tail -3 gpurun_out/r4_pytest_gpu.log
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bench_default2.log 2>&1; echo "rc=$?" >> gpurun_out/r4_bench_default2.log)
python - <<'PY'
import json
for l in open('gpurun_out/r4_bench_default2.log'):
    if l.startswith('{'):
        d=json.loads(l); o=d['other_workloads']
        print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'single', d['single_grid_latency_us'], 'dropin', d.get('dropin_call_ms'), d.get('secondary_error'))
        print('stream_cfg4', o.get('stream_cfg4')); print('xtc_cfg4', o.get('xtc_cfg4')); print('cfg4', o['cfg4']['ms_per_step'])
PY
tail -1 gpurun_out/r4_bench_default2.log
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --min-seconds 1 > gpurun_out/r4_bench_torchrun1.log 2>&1; echo "rc=$?" >> gpurun_out/r4_bench_torchrun1.log)
python - <<'PY'
import json
for l in open('gpurun_out/r4_bench_torchrun1.log'):
    if l.startswith('{'):
        d=json.loads(l); print('torchrun1: value', d['value'], 'gather_ms', d.get('gather_ms'), 'overlapped_extra', d.get('gather_overlapped_extra_ms'), 'exchange', d.get('gather_exchange'), 'err', d.get('gather_error'), 'alive', d.get('ranks_alive'))
PY
tail -2 gpurun_out/r4_bench_torchrun1.log | cut -c1-300
