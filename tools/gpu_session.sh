# round 6, session 70: the measured choice for selfdist (triangular row kernel: >= 700 atoms up to 32 frames, >= 1 500 at any frame count): whole GPU tier, probes, sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s70_tests.txt
timeout 600 python tools/dist_few_frames_probe.py 2>&1 | grep -v amdgpu > gpurun_out/dist_few_frames_probe.txt; grep self gpurun_out/dist_few_frames_probe.txt | cut -c1-150
timeout 1200 python tests/sweep_gpu_dist.py 20000 4000 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/sweep_dist_few.txt
(timeout 600 python bench.py --workload dist --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/dist_line.json)
python -c "
import json; d = json.load(open('gpurun_out/dist_line.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac']); print({k: (v.get('roofline', {}).get('frac') if isinstance(v, dict) else None) for k, v in d.items() if isinstance(v, dict)})"
