# round 6, session 74: long random parity sweeps on the FINAL build (library 721c8f643aa04dd8)
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python tests/sweep_gpu_random.py 60000 4000; timeout 1500 python tests/sweep_gpu_dist.py 60000 12000; timeout 1200 python tests/sweep_gpu_reduction.py 60000 6000; timeout 1200 python tests/sweep_gpu_contacts.py 60000 3000; timeout 900 python tests/sweep_gpu_topology.py 60000 8000) 2>&1 | grep -v amdgpu | tee gpurun_out/random_sweeps_long.txt | tail -12
