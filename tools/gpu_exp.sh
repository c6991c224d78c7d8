mkdir -p gpurun_out; export TMPDIR=/tmp
for wl in cfg2 cfg3 cfg5 cfg4; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > gpurun_out/bench_$wl.log 2>&1
done
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
