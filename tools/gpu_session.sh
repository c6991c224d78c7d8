# round 6, session 73: short-row calls of few frames through the row kernel the other way round: GPU tier, probe, sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s73_tests.txt
timeout 600 python tools/dist_short_rows_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/dist_short_rows_probe.txt | cut -c1-230
timeout 1200 python tests/sweep_gpu_dist.py 50000 4000 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/sweep_dist_few.txt
