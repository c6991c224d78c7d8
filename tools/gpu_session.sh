# round 6, session 63: random topology calls with wide atoms: split fix-up == inside k_tail == plain call, bit for bit
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tests/sweep_gpu_topology.py 0 6000 2>&1 | grep -v amdgpu | tail -15 | tee gpurun_out/sweep_topology.txt
