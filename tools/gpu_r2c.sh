# tools/gpu_r2c.sh -- A/B of the binning cell size (half-cutoff cells + packed candidate chunks vs cutoff-sized cells)
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
for mode in 0 1; do
  (MKAMD_COARSE_CELLS=$mode timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_cells$mode.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cells$mode.log)
  (MKAMD_COARSE_CELLS=$mode timeout 600 python bench.py --no-cpu-baseline --no-extra --no-pipeline > gpurun_out/bench_cells${mode}_nopipe.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cells${mode}_nopipe.log)
done
tail -4 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
for f in ("bench_cells0","bench_cells1","bench_cells0_nopipe","bench_cells1_nopipe"):
    for l in open(f'gpurun_out/{f}.log'):
        if l.startswith('{'):
            d=json.loads(l)
            print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_ms'], d.get('single_grid_latency_us'))
            for k,v in d.get('other_workloads',{}).items():
                print('   ',k, v['value'], v['ms_per_step'], v['roofline']['frac'], v['roofline']['kernel_avg_ms'])
PY
