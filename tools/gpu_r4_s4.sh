mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_gpu_distance.py -q -x 2>&1 | tail -3)
(timeout 300 python bench.py --workload dist --steps 20 --warmup 3 > gpurun_out/r4_bench_dist2.log 2>&1; echo "rc=$?" >> gpurun_out/r4_bench_dist2.log)
tail -2 gpurun_out/r4_bench_dist2.log
(timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_default.log 2>&1; echo "rc=$?" >> gpurun_out/r4_bench_default.log)
tail -2 gpurun_out/r4_bench_default.log
