# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 9: the frame kernel with four frames per block -- GPU distance tests, the shapes probe, the dist bench (clocks warmed)
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x > gpurun_out/pytest_gpu_dist.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_dist.log)
tail -4 gpurun_out/pytest_gpu_dist.log
(timeout 300 python tools/dist_shapes_probe.py > gpurun_out/dist_shapes_probe4.txt 2>&1); grep -v "avoid=3" gpurun_out/dist_shapes_probe4.txt | head -24
(timeout 300 python bench.py --workload dist --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/bench_dist.log 2>&1); tail -c 2500 gpurun_out/bench_dist.log
