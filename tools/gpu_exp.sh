mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; }
for rep in 1 2 3; do
EXTRA_ARGS="" run cfg2_new A=1
EXTRA_ARGS="" run cfg2_head MKAMD_LIB=$R/.variants/lib_head.so
done
EXTRA_ARGS="--workload cfg4" run cfg4_new A=1
EXTRA_ARGS="--workload cfg4" run cfg4_head MKAMD_LIB=$R/.variants/lib_head.so
python tools/bench_prepass_heavy.py 48 | tail -1
MKAMD_LIB=$R/.variants/lib_head.so python tools/bench_prepass_heavy.py 48 | tail -1
