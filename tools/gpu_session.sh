# round 6, session 57: the group reductions of few frames (k_dist_reduction_few): GPU test, crossover probe
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_distance.py -x -q -m gpu -k "few_frames or reduction" 2>&1 | tail -5 | tee gpurun_out/s57_tests.txt
timeout 900 python tools/reduction_few_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/reduction_few_probe.txt
