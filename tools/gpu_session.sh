# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 7: the cheaper minimum-image test + the dispatch rule -- GPU tests, shapes probe, dist bench leg (with its new legs), sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
tail -4 gpurun_out/pytest_gpu.log
(timeout 300 python tools/dist_shapes_probe.py > gpurun_out/dist_shapes_probe3.txt 2>&1); grep -v "avoid=3" gpurun_out/dist_shapes_probe3.txt
(timeout 300 python bench.py --workload dist --steps 20 --warmup 3 > gpurun_out/bench_dist.log 2>&1); tail -c 3000 gpurun_out/bench_dist.log
(timeout 400 python tests/sweep_gpu_dist.py 0 300 > gpurun_out/sweep_dist.txt 2>&1); tail -3 gpurun_out/sweep_dist.txt
