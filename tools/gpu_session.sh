# round 6, session 65: dist_trajectory at few frames takes the row kernel (lanes along the second atoms): GPU distance tests, the probe, the random sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_distance.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s65_tests.txt
timeout 600 python tools/dist_few_frames_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/dist_few_frames_probe.txt | grep "rect" | cut -c1-250
timeout 900 python tests/sweep_gpu_dist.py 5000 1500 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/sweep_dist_few.txt
