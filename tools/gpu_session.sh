# round 6, session 67: the final choice for few frames (row kernel <= 32 frames; selfdist through it <= 6 frames): whole GPU tier, the probe, the sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s67_tests.txt
timeout 600 python tools/dist_few_frames_probe.py 2>&1 | grep -v amdgpu > gpurun_out/dist_few_frames_probe.txt; grep self gpurun_out/dist_few_frames_probe.txt | cut -c1-200
timeout 900 python tests/sweep_gpu_dist.py 10000 2000 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/sweep_dist_few.txt
