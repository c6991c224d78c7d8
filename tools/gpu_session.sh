# round 6, session 52: few-frame calls: rows of the masks and counters hold just the frames: GPU distance tests, get_collisions
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_distance.py -m gpu -q -x 2>&1 | tail -3)
timeout 600 python tools/collisions_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/collisions_probe.txt
