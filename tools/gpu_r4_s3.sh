# round 4, session 3: GPU tests, the distance leg (periodic + the new non-periodic one), then the tile kernel's time map (diag builds)
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python bench.py --workload dist --steps 20 --warmup 3 > gpurun_out/r4_bench_dist.log 2>&1; echo "rc=$?" >> gpurun_out/r4_bench_dist.log)
tail -2 gpurun_out/r4_bench_dist.log
bash tools/gpu_r4_tile_ab.sh "$@"
