# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 21: more random parity on the final build: 3 000 further voxelizer configurations (automatic mode), 600 through the
# workgroup-per-item kernel, 2 000 further dist_trajectory shapes
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python tests/sweep_gpu_random.py 20000 3000; MKAMD_TILE_ITEMS=1 timeout 600 python tests/sweep_gpu_random.py 30000 600; timeout 1200 python tests/sweep_gpu_dist.py 1000 2000) > gpurun_out/random_sweeps_extra.txt 2>&1
grep -h "worst\|differ" gpurun_out/random_sweeps_extra.txt | cut -c1-200
