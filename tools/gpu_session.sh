# round 6, session 68: the new GPU test of dist_trajectory at few frames
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_distance.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/s68_tests.txt
