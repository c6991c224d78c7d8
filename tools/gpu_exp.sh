mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d.get('single_grid_latency_us'))"; }
(MKAMD_FORCE_GENERAL=1 timeout 400 python bench.py --no-cpu-baseline --no-pipeline > gpurun_out/bench_cfg2_general.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg2_general.log)
for b in 1 8 32 64 128 256 512 1024; do
EXTRA_ARGS="--batch $b" run cfg2_B$b A=1
done
for b in 256 1024 4096 8192 32768; do
EXTRA_ARGS="--workload cfg3 --batch $b" run cfg3_B$b A=1
done
