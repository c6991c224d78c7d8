mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -1
for wl in cfg3 cfg5 cfg1 cfg2; do
EXTRA_ARGS="--workload $wl" run ${wl}_new A=1
EXTRA_ARGS="--workload $wl" run ${wl}_head MKAMD_LIB=$R/.variants/lib_head.so
done
