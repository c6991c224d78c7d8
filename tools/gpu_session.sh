# round 6, session 54: random sweep of the contact lists against the oracle
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tests/sweep_gpu_contacts.py 400 2000 2>&1 | grep -v amdgpu | tail -12 | tee gpurun_out/sweep_contacts.txt
