# round 6, session 71: long random parity sweeps on the FINAL build (library 68c9201e3874b22b): voxelizer, dist_trajectory (few frames, triangular rows), reductions (few frames), contacts, topology calls with wide atoms
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python tests/sweep_gpu_random.py 30000 4000; timeout 1500 python tests/sweep_gpu_dist.py 30000 12000; timeout 1200 python tests/sweep_gpu_reduction.py 30000 6000; timeout 1200 python tests/sweep_gpu_contacts.py 30000 3000; timeout 900 python tests/sweep_gpu_topology.py 30000 8000) 2>&1 | grep -v amdgpu | tee gpurun_out/random_sweeps_long.txt | tail -12
