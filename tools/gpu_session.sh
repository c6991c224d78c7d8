# round 6, session 58: the whole GPU tier on the build with k_dist_reduction_few, packed host calls and the known-answer replay
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee gpurun_out/s58_tests.txt
