# round 6, session 56: packed host calls (host_pack.h) -- GPU test, then the PCIe-inclusive figure of the distance row
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_distance.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/s56_tests.txt
(timeout 600 python bench.py --workload dist --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/dist_line.json)
python -c "
import json; d = json.load(open('gpurun_out/dist_line.json')); print(json.dumps(d['host_call'], indent=1)); print(d['ms_per_step'], d['value'])"
