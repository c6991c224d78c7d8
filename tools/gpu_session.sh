# round 6, session 16: ShardedDistances on the device; a random sweep of the group reductions
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -4)
(timeout 900 python tests/sweep_gpu_reduction.py 0 400 2>&1 | grep -v amdgpu | tail -5) | tee gpurun_out/s16_reduction_sweep.txt
