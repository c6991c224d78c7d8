# round 6, session 59: k_dist_reduction_few in packed arithmetic: GPU tests, crossover probe
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_distance.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/s59_tests.txt
timeout 900 python tools/reduction_few_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/reduction_few_probe.txt
(timeout 600 python bench.py --workload dist --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/dist_line.json)
python -c "
import json; d = json.load(open('gpurun_out/dist_line.json')); print(json.dumps(d['host_call'], indent=1)); print(json.dumps(d['reduction']['one_frame'])); print(d['ms_per_step'], d['value'])"
