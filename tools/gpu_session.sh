# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 6, session 1: the new closest-atom group reduction kernel, the device-resident entry points and the reference-held
# MetricDistance projections on the hardware; then the whole GPU tier; then the dist bench line with its new legs
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_distance.py -m gpu -q -x -s 2>&1 | tail -15) > gpurun_out/s1_dist_tests.txt 2>&1
cat gpurun_out/s1_dist_tests.txt
(timeout 600 python bench.py --workload dist --steps 20 --warmup 3 > gpurun_out/s1_bench_dist.log 2>&1; echo "rc=$?" >> gpurun_out/s1_bench_dist.log)
tail -c 6000 gpurun_out/s1_bench_dist.log
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > gpurun_out/s1_all_tests.txt 2>&1
cat gpurun_out/s1_all_tests.txt
